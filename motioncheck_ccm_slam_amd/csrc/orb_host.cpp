// orb_host.cpp -- host side of the extractor: constructor tables, pyramid/cell geometry,
// device buffers, launch sequence and the C ABI entry points (include/ccm_hot.h).
// Reference: cslam/src/ORBextractor.cpp (cited per function).
#include "ccm_internal.h"
#include "orb_types.h"
#include <cmath>
#include <cfloat>
#include <algorithm>

extern "C" hipError_t orb_upload_pattern();
size_t orb_octree_lds_bytes(int list_cap);
void orb_launch_resize(hipStream_t, const OrbGeom&, int level, int dw, int dh, int nframes);
void orb_launch_score(hipStream_t, const OrbGeom&, int ntiles, int nframes);
void orb_launch_nms(hipStream_t, const OrbGeom&, const OrbCell*, int ncells, int nframes, unsigned* slots, int* cell_count);
void orb_launch_fast_cells(hipStream_t, const OrbGeom&, const OrbCell*, const OrbBand*, int nbands, int nframes, size_t lds_bytes,
                           int surv_cap, unsigned* slots, int* cell_count);
size_t orb_fast_cells_lds(int pitch, int bh, int surv_cap);
void orb_launch_octree(hipStream_t, const OrbGeom&, const OrbCell*, int nlevels, int nframes, int list_cap,
                       const unsigned* slots, const int* cell_count, unsigned* keysA, unsigned* keysB,
                       unsigned* out, int* out_count, int* status);
void orb_launch_orient_desc(hipStream_t, const OrbGeom&, int out_per_frame, int nframes, const unsigned* sel,
                            const int* sel_count, ccm_keypoint* kps, uint8_t* desc, int* counts, int max_per_image,
                            int* status);

namespace {

inline int cv_round(float v) { return (int)lrintf(v); }       // cvRound: ties to even (default FP mode)
inline int cv_round(double v) { return (int)lrint(v); }

struct OrbTables {
    float scale[ORB_MAX_LEVELS], inv_scale[ORB_MAX_LEVELS], sigma2[ORB_MAX_LEVELS], inv_sigma2[ORB_MAX_LEVELS];
    int nfeat[ORB_MAX_LEVELS];
    int umax[16];
};

// ORBextractor::ORBextractor, ORBextractor.cpp:579-639
int make_tables(const ccm_orb_params* p, OrbTables* t)
{
    if (!p || p->nlevels < 1 || p->nlevels > ORB_MAX_LEVELS || p->nfeatures < 0 || !(p->scale_factor > 1.0f)) return CCM_E_ARG;
    const int n = p->nlevels;
    t->scale[0] = 1.0f; t->sigma2[0] = 1.0f;
    for (int i = 1; i < n; i++) {
        t->scale[i] = t->scale[i - 1] * p->scale_factor;
        t->sigma2[i] = t->scale[i] * t->scale[i];
    }
    for (int i = 0; i < n; i++) {
        t->inv_scale[i] = 1.0f / t->scale[i];
        t->inv_sigma2[i] = 1.0f / t->sigma2[i];
    }
    const float factor = 1.0f / p->scale_factor;
    float desired = p->nfeatures * (1 - factor) / (1 - (float)std::pow((double)factor, (double)n));
    int sum = 0;
    for (int l = 0; l < n - 1; l++) {
        t->nfeat[l] = cv_round(desired);
        sum += t->nfeat[l];
        desired *= factor;
    }
    t->nfeat[n - 1] = std::max(p->nfeatures - sum, 0);
    // circular-patch half widths (:623-638)
    const int vmax = (int)std::floor(ORB_HALF_PATCH * std::sqrt(2.f) / 2 + 1);
    const int vmin = (int)std::ceil(ORB_HALF_PATCH * std::sqrt(2.f) / 2);
    const double hp2 = ORB_HALF_PATCH * ORB_HALF_PATCH;
    for (int v = 0; v < 16; v++) t->umax[v] = 0;
    for (int v = 0; v <= vmax; ++v) t->umax[v] = cv_round(std::sqrt(hp2 - v * v));
    for (int v = ORB_HALF_PATCH, v0 = 0; v >= vmin; --v) {
        while (t->umax[v0] == t->umax[v0 + 1]) ++v0;
        t->umax[v] = v0;
        ++v0;
    }
    return CCM_OK;
}

// cv::resize(INTER_LINEAR) coefficient tables for one axis
void linear_tables(int ssize, int dsize, std::vector<int>& ofs, std::vector<short>& ab)
{
    ofs.resize(dsize); ab.resize(2 * (size_t)dsize);
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float fx = (float)((d + 0.5) * scale - 0.5);
        int sx = (int)std::floor(fx);
        fx -= sx;
        if (sx < 0) { sx = 0; fx = 0.f; }
        if (sx >= ssize - 1) { sx = ssize - 1; fx = 0.f; }
        ofs[d] = sx;
        ab[2 * d] = (short)cv_round((1.f - fx) * 2048.f);
        ab[2 * d + 1] = (short)cv_round(fx * 2048.f);
    }
}

}  // namespace

struct OrbState {
    // cache key
    ccm_orb_params par{}; int w = 0, h = 0, nframes = 0, max_per_image = 0;
    bool valid = false, pattern_up = false;
    OrbTables tab{};
    OrbGeom geom{};
    std::vector<OrbCell> cells;
    std::vector<OrbBand> bands; size_t band_lds = 0; bool fused = true; int surv_cap = 1024; hipStream_t copy_stream = nullptr; hipEvent_t copy_done = nullptr;
    hipStream_t down_stream = nullptr; hipEvent_t chunk_done = nullptr;      // results of a chunk go down while the next chunk is extracted
    DevBuf geom_dev, cells_dev, tables_dev, bands_dev;
    DevBuf pyr, smap, slots, cell_count, keysA, keysB, sel, sel_count, status;
    DevBuf img0;                   // staging for the host-pointer entry point
    DevBuf kps, desc, counts;      // results
    size_t pyr_off[ORB_MAX_LEVELS]{}, smap_off[ORB_MAX_LEVELS]{};
    const uint8_t* last_img = nullptr; int last_stride = 0; size_t last_image_stride = 0;
    bool have_result = false;
};

void orb_state_free(OrbState* s)
{
    if (!s) return;
    DevBuf* all[] = { &s->geom_dev, &s->cells_dev, &s->tables_dev, &s->bands_dev, &s->pyr, &s->smap, &s->slots, &s->cell_count,
                      &s->keysA, &s->keysB, &s->sel, &s->sel_count, &s->status, &s->img0, &s->kps, &s->desc, &s->counts };
    for (DevBuf* b : all) b->release();
    if (s->copy_done) (void)hipEventDestroy(s->copy_done);
    if (s->chunk_done) (void)hipEventDestroy(s->chunk_done);
    delete s;
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Build geometry + device buffers for (params, w, h, nframes).  Level 0's image pointer is patched per call.
static int orb_prepare(ccm_ctx* c, const ccm_orb_params* p, int w, int h, int nframes, int max_per_image)
{
    if (!c->orb) c->orb = new OrbState();
    OrbState& S = *c->orb;
    const bool same = S.valid && std::memcmp(&S.par, p, sizeof *p) == 0 && S.w == w && S.h == h &&
                      S.nframes == nframes && S.max_per_image == max_per_image;
    if (same) return CCM_OK;
    S.valid = false; S.have_result = false;
    int rc = make_tables(p, &S.tab);
    if (rc) return ccm_fail(c, rc, "bad ORB parameters");
    if (p->ini_th_fast < 0 || p->ini_th_fast > 255 || p->min_th_fast < 0 || p->min_th_fast > 255)
        return ccm_fail(c, CCM_E_ARG, "FAST thresholds must be in [0,255]");
    OrbGeom& G = S.geom;
    std::memset(&G, 0, sizeof G);
    G.nlevels = p->nlevels; G.ini_th = p->ini_th_fast; G.min_th = p->min_th_fast;
    for (int i = 0; i < 16; i++) G.umax[i] = S.tab.umax[i];
    S.cells.clear();
    std::vector<int> tab_i; std::vector<short> tab_s;          // concatenated resize tables
    struct TabOff { size_t xofs, yofs, xab, yab; } toff[ORB_MAX_LEVELS]{};
    size_t pyr_bytes = 0, smap_bytes = 0;
    int slot_acc = 0, key_acc = 0, out_acc = 0, tile_acc = 0, max_quota = 0;
    for (int l = 0; l < p->nlevels; l++) {
        OrbLevel& L = G.lv[l];
        L.w = cv_round((float)w * S.tab.inv_scale[l]);          // :1284-1285
        L.h = cv_round((float)h * S.tab.inv_scale[l]);
        if (L.w < 2 * ORB_EDGE + 8 || L.h < 2 * ORB_EDGE + 8 || L.w > 4096 + 2 * ORB_BORDER || L.h > 4096 + 2 * ORB_BORDER)
            return ccm_fail(c, CCM_E_ARG, "level %d is %dx%d: unsupported image size", l, L.w, L.h);
        L.scale = S.tab.scale[l];
        L.kp_size = (float)(int)(31 * S.tab.scale[l]);          // :1006
        L.quota = S.tab.nfeat[l];
        max_quota = std::max(max_quota, L.quota);
        L.spitch = (int)align_up(L.w, 64);
        L.splane = (long long)L.spitch * L.h;
        S.smap_off[l] = smap_bytes; smap_bytes += align_up((size_t)L.splane * nframes, 256);
        if (l > 0) {
            L.pitch = (int)align_up(L.w, 64);
            L.plane = (long long)L.pitch * L.h;
            S.pyr_off[l] = pyr_bytes; pyr_bytes += align_up((size_t)L.plane * nframes, 256);
            std::vector<int> xo, yo; std::vector<short> xa, ya;
            linear_tables(G.lv[l - 1].w, L.w, xo, xa);
            linear_tables(G.lv[l - 1].h, L.h, yo, ya);
            toff[l].xofs = tab_i.size(); tab_i.insert(tab_i.end(), xo.begin(), xo.end());
            toff[l].yofs = tab_i.size(); tab_i.insert(tab_i.end(), yo.begin(), yo.end());
            toff[l].xab = tab_s.size(); tab_s.insert(tab_s.end(), xa.begin(), xa.end());
            toff[l].yab = tab_s.size(); tab_s.insert(tab_s.end(), ya.begin(), ya.end());
        }
        // FAST cells (:941-974)
        const int minBX = ORB_BORDER, minBY = ORB_BORDER, maxBX = L.w - ORB_EDGE + 3, maxBY = L.h - ORB_EDGE + 3;
        const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
        const int nCols = (int)(width / 30.f), nRows = (int)(height / 30.f);
        L.bw = maxBX - minBX; L.bh = maxBY - minBY;
        L.cell_first = (int)S.cells.size();
        int key_cap = 0;
        if (nCols >= 1 && nRows >= 1) {
            const int wCell = (int)std::ceil(width / nCols), hCell = (int)std::ceil(height / nRows);
            for (int i = 0; i < nRows; i++) {
                const float iniY = (float)(minBY + i * hCell);
                float maxY = iniY + hCell + 6;
                if (iniY >= maxBY - 3) continue;
                if (maxY > maxBY) maxY = (float)maxBY;
                for (int j = 0; j < nCols; j++) {
                    const float iniX = (float)(minBX + j * wCell);
                    float maxX = iniX + wCell + 6;
                    if (iniX >= maxBX - 6) continue;
                    if (maxX > maxBX) maxX = (float)maxBX;
                    OrbCell cell{};
                    cell.level = (short)l; cell.x0 = (short)iniX; cell.y0 = (short)iniY;
                    cell.cw = (short)((int)maxX - (int)iniX); cell.ch = (short)((int)maxY - (int)iniY);
                    if (cell.cw > 65 || cell.ch > 65)
                        return ccm_fail(c, CCM_E_ARG, "FAST cell %dx%d exceeds the kernel tile", cell.cw, cell.ch);
                    const int rw = std::max(cell.cw - 6, 0), rh = std::max(cell.ch - 6, 0);
                    // strict 3x3 maxima cannot touch, even diagonally: at most ceil(rw/2)*ceil(rh/2)
                    cell.slot_cap = ((rw + 1) / 2) * ((rh + 1) / 2);
                    cell.slot_first = slot_acc;
                    slot_acc += cell.slot_cap; key_cap += cell.slot_cap;
                    S.cells.push_back(cell);
                }
            }
        }
        L.ncells = (int)S.cells.size() - L.cell_first;
        // DistributeOctTree roots (:711-713)
        L.roots = (int)std::round((float)(maxBX - minBX) / (maxBY - minBY));
        if (L.roots < 1 || L.roots > ORB_MAX_ROOTS)
            return ccm_fail(c, CCM_E_ARG, "level %d aspect ratio gives %d quadtree roots (1..%d supported)", l, L.roots, ORB_MAX_ROOTS);
        L.hx = (float)(maxBX - minBX) / L.roots;
        L.key_first = key_acc; L.key_cap = std::max(key_cap, 1); key_acc += L.key_cap;
        L.out_cap = std::max(L.quota + 4, 4 * L.roots);
        L.out_first = out_acc; out_acc += L.out_cap;
        L.tiles_x = (L.w + 63) / 64; L.tiles_y = (L.h + 63) / 64;      // ST_W x ST_H of k_fast_score
        L.tile_first = tile_acc; tile_acc += L.tiles_x * L.tiles_y;
    }
    // bands of the fused FAST kernel: runs of consecutive cells of one cell row whose LDS tile stays small
    S.bands.clear(); S.band_lds = 0;
    static const bool want_fused = !(getenv("CCM_ORB_FUSED") && atoi(getenv("CCM_ORB_FUSED")) == 0);
    S.fused = want_fused;
    static const int pcap = getenv("CCM_FC_PCAP") ? std::min(atoi(getenv("CCM_FC_PCAP")), 256) : 144;   // measured: 112 0.576, 128 0.558, 144 0.532, 160 0.574 ms      // pitch <= 256: one dword per lane
    static const int scap = getenv("CCM_FC_SURV") ? atoi(getenv("CCM_FC_SURV")) : 3072;      // two row blocks per 30-row band (measured best)
    S.surv_cap = scap;
    for (size_t a = 0; a < S.cells.size();) {
        size_t b = a;
        const OrbCell& c0 = S.cells[a];
        const int xa = (c0.x0 - 4) & ~3;
        int pitch = 0;
        while (b < S.cells.size() && S.cells[b].level == c0.level && S.cells[b].y0 == c0.y0) {
            const int p2 = (int)align_up((size_t)(S.cells[b].x0 + S.cells[b].cw + 4 - xa), 16);   // 16: the tile is staged 16 bytes per lane
            if (b > a && (p2 > pcap || b - a >= ORB_BAND_CELLS)) break;                             // (the band record carries its cells: at most ORB_BAND_CELLS)
            pitch = p2; b++;
        }
        OrbBand band{};
        band.cell_first = (int)a; band.ncells = (int)(b - a); band.level = c0.level; band.xa = (short)xa; band.y0 = (short)c0.y0;
        band.pitch = (short)pitch; band.bh = (short)c0.ch;
        {
            const OrbCell& cl = S.cells[b - 1];
            const int PW = pitch / 4, PQ = std::max(pitch / 16, 1);
            band.c_lo = (short)(c0.x0 + 3 - xa); band.c_hi = (short)(cl.x0 + cl.cw - 3 - xa);
            band.dw_lo = (short)std::max(1, band.c_lo >> 2); band.dw_hi = (short)std::min(PW - 2, (band.c_hi - 1) >> 2);
            band.pw_inv = (1u << 20) / (unsigned)PW + 1u; band.pq_inv = (1u << 20) / (unsigned)PQ + 1u;
            band.w = G.lv[c0.level].w; band.h = G.lv[c0.level].h;
            for (size_t k = a; k < b; k++) {
                const OrbCell& ck = S.cells[k];
                band.clo[k - a] = (short)(ck.x0 + 3 - xa); band.cwd[k - a] = (short)(ck.cw - 6);
                band.slot_first[k - a] = ck.slot_first; band.slot_cap[k - a] = ck.slot_cap;
            }
        }
        for (size_t k = a; k < b; k++) if (S.cells[k].ch != c0.ch) S.fused = false;      // cannot happen: one row, one height
        S.surv_cap = std::max(S.surv_cap, pitch);                 // at least one row per block
        S.bands.push_back(band);
        a = b;
    }
    for (const OrbBand& b : S.bands) S.band_lds = std::max(S.band_lds, orb_fast_cells_lds(b.pitch, b.bh, S.surv_cap));
    for (const OrbBand& b : S.bands) if (b.pitch > 256) S.fused = false;    // a single cell wider than the tile: two-kernel path
    if (S.band_lds > 60 * 1024) S.fused = false;                 // very large cells: keep the two-kernel path
    G.ncells = (int)S.cells.size(); G.ntiles = tile_acc;
    G.slots_per_frame = std::max(slot_acc, 1); G.keys_per_frame = key_acc; G.out_per_frame = out_acc;
    int list_cap = 0;
    for (int l = 0; l < p->nlevels; l++) list_cap = std::max(list_cap, G.lv[l].out_cap);
    G.list_cap = (int)align_up(list_cap + 4, 16);
    if (orb_octree_lds_bytes(G.list_cap) > 150 * 1024)
        return ccm_fail(c, CCM_E_ARG, "nfeatures too large for the quadtree kernel's LDS (list cap %d)", G.list_cap);
    if (max_per_image < 1) return ccm_fail(c, CCM_E_ARG, "max_per_image must be positive");

    // device buffers
    CCM_RESERVE(c, S.pyr, std::max<size_t>(pyr_bytes, 256));
    CCM_RESERVE(c, S.smap, smap_bytes);
    CCM_RESERVE(c, S.slots, (size_t)G.slots_per_frame * nframes * 4);
    CCM_RESERVE(c, S.cell_count, (size_t)std::max(G.ncells, 1) * nframes * 4);
    CCM_RESERVE(c, S.keysA, (size_t)G.keys_per_frame * nframes * 4);
    CCM_RESERVE(c, S.keysB, (size_t)G.keys_per_frame * nframes * 4);
    CCM_RESERVE(c, S.sel, (size_t)G.out_per_frame * nframes * 4);
    CCM_RESERVE(c, S.sel_count, (size_t)p->nlevels * nframes * 4);
    CCM_RESERVE(c, S.status, 256);
    CCM_RESERVE(c, S.kps, (size_t)nframes * max_per_image * sizeof(ccm_keypoint));
    CCM_RESERVE(c, S.desc, (size_t)nframes * max_per_image * 32);
    CCM_RESERVE(c, S.counts, (size_t)nframes * 4);
    const size_t tab_bytes = align_up(tab_i.size() * 4, 16) + tab_s.size() * 2;
    CCM_RESERVE(c, S.tables_dev, std::max<size_t>(tab_bytes, 16));
    CCM_RESERVE(c, S.cells_dev, std::max<size_t>(S.cells.size(), 1) * sizeof(OrbCell));
    CCM_RESERVE(c, S.bands_dev, std::max<size_t>(S.bands.size(), 1) * sizeof(OrbBand));
    CCM_RESERVE(c, S.geom_dev, sizeof(OrbGeom));
    char* tb = S.tables_dev.as<char>();
    const size_t s_base = align_up(tab_i.size() * 4, 16);
    for (int l = 0; l < p->nlevels; l++) {
        OrbLevel& L = G.lv[l];
        L.smap = S.smap.as<uint8_t>() + S.smap_off[l];
        if (l > 0) {
            L.img = S.pyr.as<uint8_t>() + S.pyr_off[l];
            L.xofs = reinterpret_cast<const int*>(tb) + toff[l].xofs;
            L.yofs = reinterpret_cast<const int*>(tb) + toff[l].yofs;
            L.xab = reinterpret_cast<const short*>(tb + s_base) + toff[l].xab;
            L.yab = reinterpret_cast<const short*>(tb + s_base) + toff[l].yab;
        }
    }
    if (!tab_i.empty()) CCM_HIP(c, hipMemcpyAsync(tb, tab_i.data(), tab_i.size() * 4, hipMemcpyHostToDevice, c->stream));
    if (!tab_s.empty()) CCM_HIP(c, hipMemcpyAsync(tb + s_base, tab_s.data(), tab_s.size() * 2, hipMemcpyHostToDevice, c->stream));
    if (!S.cells.empty())
        CCM_HIP(c, hipMemcpyAsync(S.cells_dev.p, S.cells.data(), S.cells.size() * sizeof(OrbCell), hipMemcpyHostToDevice, c->stream));
    for (OrbBand& b : S.bands) {                       // the level images' device addresses are known now
        const OrbLevel& L = G.lv[b.level];
        b.img = b.level > 0 ? L.img : nullptr; b.plane = L.plane; b.lpitch = L.pitch;
    }
    if (!S.bands.empty())
        CCM_HIP(c, hipMemcpyAsync(S.bands_dev.p, S.bands.data(), S.bands.size() * sizeof(OrbBand), hipMemcpyHostToDevice, c->stream));
    CCM_HIP(c, hipStreamSynchronize(c->stream));     // the host vectors above go out of scope
    if (!S.pattern_up) { CCM_HIP(c, orb_upload_pattern()); S.pattern_up = true; }
    S.par = *p; S.w = w; S.h = h; S.nframes = nframes; S.max_per_image = max_per_image;
    S.valid = true;
    S.last_img = nullptr;
    return CCM_OK;
}

// Runs the pipeline on frames [frame0, frame0 + nrun) of the prepared batch (all of it by default).
static int orb_run(ccm_ctx* c, const uint8_t* img_dev, int stride, size_t image_stride, int frame0 = 0, int nrun = -1)
{
    OrbState& S = *c->orb;
    OrbGeom& G = S.geom;
    // the geometry table (incl. level 0's image pointer) travels by value in the kernel arguments
    G.lv[0].img = img_dev; G.lv[0].pitch = stride; G.lv[0].plane = (long long)image_stride;
    G.frame0 = frame0;
    if (nrun < 0) nrun = S.nframes - frame0;
    const OrbGeom& gd = G;
    const OrbCell* cd = S.cells_dev.as<OrbCell>();
    hipStream_t st = c->stream;
    if (frame0 == 0) CCM_HIP(c, hipMemsetAsync(S.status.p, 0, 4, st));
    for (int l = 1; l < G.nlevels; l++) { ProfScope ps(c, CCM_PROF_RESIZE); orb_launch_resize(st, gd, l, G.lv[l].w, G.lv[l].h, nrun); }
    if (S.fused && !S.bands.empty()) {
        ProfScope ps(c, CCM_PROF_FAST_SCORE);
        orb_launch_fast_cells(st, gd, cd, S.bands_dev.as<OrbBand>(), (int)S.bands.size(), nrun, S.band_lds, S.surv_cap, S.slots.as<unsigned>(), S.cell_count.as<int>());
    } else {
        { ProfScope ps(c, CCM_PROF_FAST_SCORE); orb_launch_score(st, gd, G.ntiles, nrun); }
        if (G.ncells > 0) { ProfScope ps(c, CCM_PROF_CELL_NMS); orb_launch_nms(st, gd, cd, G.ncells, nrun, S.slots.as<unsigned>(), S.cell_count.as<int>()); }
    }
    { ProfScope ps(c, CCM_PROF_OCTREE);
    orb_launch_octree(st, gd, cd, G.nlevels, nrun, G.list_cap, S.slots.as<unsigned>(), S.cell_count.as<int>(),
                      S.keysA.as<unsigned>(), S.keysB.as<unsigned>(), S.sel.as<unsigned>(), S.sel_count.as<int>(),
                      S.status.as<int>()); }
    { ProfScope ps(c, CCM_PROF_ORIENT_DESC);
    orb_launch_orient_desc(st, gd, G.out_per_frame, nrun, S.sel.as<unsigned>(), S.sel_count.as<int>(),
                           S.kps.as<ccm_keypoint>(), S.desc.as<uint8_t>(), S.counts.as<int>(), S.max_per_image,
                           S.status.as<int>()); }
    CCM_HIP(c, hipGetLastError());
    S.have_result = true;
    return CCM_OK;
}

static int orb_check_status(ccm_ctx* c)
{
    OrbState& S = *c->orb;
    int st = 0;
    CCM_HIP(c, hipMemcpyAsync(&st, S.status.p, 4, hipMemcpyDeviceToHost, c->stream));
    CCM_HIP(c, hipStreamSynchronize(c->stream));
    if (st & 8) return ccm_fail(c, CCM_E_CAPACITY, "an image produced more than max_per_image keypoints");
    if (st) return ccm_fail(c, CCM_E_DEVICE, "extractor kernel status 0x%x (1 key overflow, 2 quadtree guard, 4 list overflow)", st);
    return CCM_OK;
}

extern "C" {

int ccm_orb_tables(const ccm_orb_params* p, float* scale, float* inv_scale, float* sigma2, float* inv_sigma2,
                   int32_t* features_per_level, int32_t* umax)
{
    OrbTables t;
    int rc = make_tables(p, &t);
    if (rc) return rc;
    for (int i = 0; i < p->nlevels; i++) {
        if (scale) scale[i] = t.scale[i];
        if (inv_scale) inv_scale[i] = t.inv_scale[i];
        if (sigma2) sigma2[i] = t.sigma2[i];
        if (inv_sigma2) inv_sigma2[i] = t.inv_sigma2[i];
        if (features_per_level) features_per_level[i] = t.nfeat[i];
    }
    if (umax) for (int i = 0; i < 16; i++) umax[i] = t.umax[i];
    return CCM_OK;
}

int ccm_orb_level_sizes(const ccm_orb_params* p, int w, int h, int32_t* level_w, int32_t* level_h)
{
    OrbTables t;
    int rc = make_tables(p, &t);
    if (rc) return rc;
    if (!level_w || !level_h) return CCM_E_ARG;
    for (int l = 0; l < p->nlevels; l++) {
        level_w[l] = cv_round((float)w * t.inv_scale[l]);
        level_h[l] = cv_round((float)h * t.inv_scale[l]);
    }
    return CCM_OK;
}

int ccm_orb_extract_dev(ccm_ctx* c, const ccm_orb_params* p, const uint8_t* img_dev, int w, int h, int stride,
                        size_t image_stride, int n_images, int max_per_image)
{
    RoctxRange roctx_("ccm_orb_extract_dev");
    if (!c || !p) return CCM_E_ARG;
    if (n_images == 0 || w == 0 || h == 0) return CCM_OK;         // empty image: silent return (:1219-1220)
    if (!img_dev || w < 0 || h < 0 || n_images < 0 || stride < w || (n_images > 1 && image_stride < (size_t)stride * h))
        return ccm_fail(c, CCM_E_ARG, "bad image arguments");
    CCM_HIP(c, hipSetDevice(c->device));
    int rc = orb_prepare(c, p, w, h, n_images, max_per_image);
    if (rc) return rc;
    return orb_run(c, img_dev, stride, image_stride);
}

int ccm_orb_fetch(ccm_ctx* c, ccm_keypoint* kps, uint8_t* desc, int32_t* counts)
{
    RoctxRange roctx_("ccm_orb_fetch");
    if (!c || !c->orb || !c->orb->have_result) return c ? ccm_fail(c, CCM_E_STATE, "no extraction to fetch") : CCM_E_ARG;
    OrbState& S = *c->orb;
    int rc = orb_check_status(c);
    if (rc) return rc;
    if (kps) CCM_HIP(c, hipMemcpyAsync(kps, S.kps.p, (size_t)S.nframes * S.max_per_image * sizeof(ccm_keypoint), hipMemcpyDeviceToHost, c->stream));
    if (desc) CCM_HIP(c, hipMemcpyAsync(desc, S.desc.p, (size_t)S.nframes * S.max_per_image * 32, hipMemcpyDeviceToHost, c->stream));
    if (counts) CCM_HIP(c, hipMemcpyAsync(counts, S.counts.p, (size_t)S.nframes * 4, hipMemcpyDeviceToHost, c->stream));
    CCM_HIP(c, hipStreamSynchronize(c->stream));
    return CCM_OK;
}

int ccm_orb_extract(ccm_ctx* c, const ccm_orb_params* p, const uint8_t* img, int w, int h, int stride,
                    size_t image_stride, int n_images, ccm_keypoint* kps, uint8_t* desc, int32_t* counts,
                    int max_per_image)
{
    RoctxRange roctx_("ccm_orb_extract");
    if (!c || !p) return CCM_E_ARG;
    if (n_images == 0 || w == 0 || h == 0) return CCM_OK;
    if (!img || w < 0 || h < 0 || n_images < 0 || stride < w) return ccm_fail(c, CCM_E_ARG, "bad image arguments");
    CCM_HIP(c, hipSetDevice(c->device));
    if (!c->orb) c->orb = new OrbState();
    OrbState& S = *c->orb;
    // Device layout of level 0: rows padded to 64 bytes (a strided 2-D copy), or -- when the caller's rows are 16-byte multiples, which
    // is all the kernels' wide loads need -- the caller's own layout, so that a chunk of frames goes up as ONE linear copy.
    // Only for batches that go up in chunks behind the extraction: a single 752 x 480 frame is extracted 26 us faster from 64-byte
    // aligned rows (0.214 against 0.240 ms per call), which a batch hides behind its uploads.
    static const bool force_2d = getenv("CCM_ORB_UPLOAD_2D") && atoi(getenv("CCM_ORB_UPLOAD_2D")) != 0;
    static const int chunk_frames = getenv("CCM_ORB_CHUNK") ? std::max(1, atoi(getenv("CCM_ORB_CHUNK"))) : 64;
    const int n_chunks = n_images >= 2 * chunk_frames ? (n_images + chunk_frames - 1) / chunk_frames : 1;
    const bool linear = !force_2d && n_chunks > 1 && stride % 16 == 0 && image_stride % 16 == 0 && image_stride >= (size_t)stride * h;
    const int pitch = linear ? stride : (int)align_up(w, 64);
    const size_t plane = linear ? image_stride : (size_t)pitch * h;
    CCM_RESERVE(c, S.img0, plane * n_images + 64);
    int rc = orb_prepare(c, p, w, h, n_images, max_per_image);
    if (rc) return rc;
    // Host buffers: the frames go up in chunks on a copy stream while the previous chunk is being extracted (the
    // upload is ~2/3 of the whole call for 752x480 frames).  Small batches go up in one piece.
    if (n_chunks > 1 && !S.copy_stream) {
        if (!(S.copy_stream = ccm_aux_stream(c, 0))) return ccm_fail(c, CCM_E_DEVICE, "hipStreamCreate failed");      // (the context's, shared)
        CCM_HIP(c, hipEventCreateWithFlags(&S.copy_done, hipEventDisableTiming));
    }
    const bool contiguous = n_images == 1 || image_stride == (size_t)stride * h;
    // Results: with page-locked destination buffers (hipHostMalloc / ccm_host_register) each chunk's keypoints, descriptors and counts
    // go down on a stream of their own while the next chunk is extracted -- the two copy directions use different engines.  A copy to
    // pageable memory holds the calling thread until it is done, which would keep it from enqueueing the next chunk: those go down
    // in one piece at the end (ccm_orb_fetch), as before.
    auto page_locked = [](const void* q) {
        if (!q) return true;
        hipPointerAttribute_t a{};
        if (hipPointerGetAttributes(&a, q) != hipSuccess) { (void)hipGetLastError(); return false; }
        return a.type == hipMemoryTypeHost;
    };
    const bool stream_down = n_chunks > 1 && (kps || desc || counts) && page_locked(kps) && page_locked(desc) && page_locked(counts);
    if (stream_down && !S.down_stream) {
        if (!(S.down_stream = ccm_aux_stream(c, 1))) return ccm_fail(c, CCM_E_DEVICE, "hipStreamCreate failed");
        CCM_HIP(c, hipEventCreateWithFlags(&S.chunk_done, hipEventDisableTiming));
    }
    for (int ck = 0; ck < n_chunks; ck++) {
        const int f0 = (int)((long long)n_images * ck / n_chunks), f1 = (int)((long long)n_images * (ck + 1) / n_chunks);
        hipStream_t up = n_chunks > 1 ? S.copy_stream : c->stream;
        if (linear) {
            const size_t bytes = f1 < n_images ? plane * (f1 - f0) : plane * (f1 - 1 - f0) + (size_t)stride * (h - 1) + w;     // (the last image may end with its last pixel)
            CCM_HIP(c, hipMemcpyAsync(S.img0.as<char>() + plane * f0, img + image_stride * f0, bytes, hipMemcpyHostToDevice, up));
        } else if (contiguous) {
            // rows of consecutive images are consecutive in both layouts (device plane == pitch * h)
            CCM_HIP(c, hipMemcpy2DAsync(S.img0.as<char>() + plane * f0, pitch, img + image_stride * f0, stride, w, (size_t)h * (f1 - f0),
                                        hipMemcpyHostToDevice, up));
        } else {
            for (int i = f0; i < f1; i++)
                CCM_HIP(c, hipMemcpy2DAsync(S.img0.as<char>() + plane * i, pitch, img + image_stride * i, stride, w, h, hipMemcpyHostToDevice, up));
        }
        if (n_chunks > 1) {
            CCM_HIP(c, hipEventRecord(S.copy_done, up));
            CCM_HIP(c, hipStreamWaitEvent(c->stream, S.copy_done, 0));
        }
        rc = orb_run(c, S.img0.as<uint8_t>(), pitch, plane, f0, f1 - f0);
        if (rc) return rc;
        if (stream_down) {
            const size_t m = (size_t)S.max_per_image, nf = (size_t)(f1 - f0);
            CCM_HIP(c, hipEventRecord(S.chunk_done, c->stream));
            CCM_HIP(c, hipStreamWaitEvent(S.down_stream, S.chunk_done, 0));
            if (kps) CCM_HIP(c, hipMemcpyAsync(kps + f0 * m, S.kps.as<ccm_keypoint>() + f0 * m, nf * m * sizeof(ccm_keypoint), hipMemcpyDeviceToHost, S.down_stream));
            if (desc) CCM_HIP(c, hipMemcpyAsync(desc + f0 * m * 32, S.desc.as<uint8_t>() + f0 * m * 32, nf * m * 32, hipMemcpyDeviceToHost, S.down_stream));
            if (counts) CCM_HIP(c, hipMemcpyAsync(counts + f0, S.counts.as<int32_t>() + f0, nf * 4, hipMemcpyDeviceToHost, S.down_stream));
        }
    }
    if (stream_down) {
        CCM_HIP(c, hipStreamSynchronize(S.down_stream));
        return orb_check_status(c);                          // (synchronises the context's stream and reads the kernels' status word)
    }
    return ccm_orb_fetch(c, kps, desc, counts);
}

int ccm_orb_result_dev(ccm_ctx* c, const uint8_t** desc_dev, const int32_t** counts_dev, int* max_per_image)
{
    if (!c || !c->orb || !c->orb->have_result) return c ? ccm_fail(c, CCM_E_STATE, "no extraction yet") : CCM_E_ARG;
    if (desc_dev) *desc_dev = c->orb->desc.as<uint8_t>();
    if (counts_dev) *counts_dev = c->orb->counts.as<int32_t>();
    if (max_per_image) *max_per_image = c->orb->max_per_image;
    return CCM_OK;
}

int ccm_orb_debug_level(ccm_ctx* c, int image, int level, uint8_t* out, int out_stride)
{
    if (!c || !c->orb || !c->orb->have_result) return c ? ccm_fail(c, CCM_E_STATE, "no extraction yet") : CCM_E_ARG;
    OrbState& S = *c->orb;
    if (level < 0 || level >= S.geom.nlevels || image < 0 || image >= S.nframes || !out) return ccm_fail(c, CCM_E_ARG, "bad level/image");
    const OrbLevel& L = S.geom.lv[level];
    CCM_HIP(c, hipMemcpy2DAsync(out, out_stride, L.img + (long long)image * L.plane, L.pitch, L.w, L.h, hipMemcpyDeviceToHost, c->stream));
    CCM_HIP(c, hipStreamSynchronize(c->stream));
    return CCM_OK;
}

int ccm_orb_debug_candidates(ccm_ctx* c, int image, int level, int32_t* xy, int32_t* score, int max)
{
    if (!c || !c->orb || !c->orb->have_result) return c ? ccm_fail(c, CCM_E_STATE, "no extraction yet") : CCM_E_ARG;
    OrbState& S = *c->orb;
    if (level < 0 || level >= S.geom.nlevels || image < 0 || image >= S.nframes) return ccm_fail(c, CCM_E_ARG, "bad level/image");
    const OrbLevel& L = S.geom.lv[level];
    std::vector<int> cnt(std::max(S.geom.ncells, 1));
    std::vector<unsigned> sl(S.geom.slots_per_frame);
    CCM_HIP(c, hipMemcpyAsync(cnt.data(), S.cell_count.as<int>() + (size_t)image * S.geom.ncells, (size_t)S.geom.ncells * 4, hipMemcpyDeviceToHost, c->stream));
    CCM_HIP(c, hipMemcpyAsync(sl.data(), S.slots.as<unsigned>() + (size_t)image * S.geom.slots_per_frame, sl.size() * 4, hipMemcpyDeviceToHost, c->stream));
    CCM_HIP(c, hipStreamSynchronize(c->stream));
    int n = 0;
    for (int ci = L.cell_first; ci < L.cell_first + L.ncells; ci++) {
        const OrbCell& cell = S.cells[ci];
        for (int k = 0; k < cnt[ci]; k++, n++) {
            if (n >= max) continue;
            const unsigned key = sl[cell.slot_first + k];
            if (xy) { xy[2 * n] = (int)(key & 0xFFFu); xy[2 * n + 1] = (int)((key >> 12) & 0xFFFu); }
            if (score) score[n] = (int)(key >> 24);
        }
    }
    return n;
}

}  // extern "C"

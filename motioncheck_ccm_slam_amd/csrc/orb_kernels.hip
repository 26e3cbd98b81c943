// orb_kernels.hip -- CDNA4 (gfx950) kernels of the ORB extractor.
//
// Pipeline per batch (all frames of the batch in every launch):
//   k_pyr_resize  x (nlevels-1)   cv::resize INTER_LINEAR 8u          ORBextractor.cpp:1293
//   k_fast_cells                  FAST-9/16 score + per-cell threshold :957-998, :978-984 (cv::FAST)
//                                 choice + 3x3 NMS, fused on cell-row bands (scores stay in LDS)
//     (k_fast_score + k_cell_nms: the same in two kernels through a score map, for cells wider than one LDS tile)
//   k_octree                      DistributeOctTree                   :707-931
//   k_orient_desc                 IC_Angle + 7x7 blur + rBRIEF        :68-95, :1259, :100-316
// Wavefront = 64 everywhere; ballots are 64-bit.
#include <hip/hip_runtime.h>
#include "orb_types.h"
#include "../../include/ccm_orb_pattern.h"
#include "../../include/ccm_sincos.h"
#include "../../include/ccm_hot.h"

#define WAVE 64

typedef unsigned short fc_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ fc_us2 fc_pk(unsigned x) { return __builtin_bit_cast(fc_us2, x); }
__device__ __forceinline__ unsigned fc_u(fc_us2 x) { return __builtin_bit_cast(unsigned, x); }

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ unsigned long long lanemask_lt() { return (1ull << lane_id()) - 1ull; }

// XCD-aware work order for 2-D grids (x = item of a frame, y = frame).  Workgroups are dealt to the 8 XCDs round-robin by their
// linear id, and each XCD has its own 4 MiB L2: with the natural order the bands / keypoints of ONE frame are spread over all
// eight, so every 128-byte line of the frame's pyramid that two neighbouring items share crosses the fabric once per XCD (the
// round-1 counters showed 2.1x the algorithmic bytes for k_fast_cells).  This bijection hands XCD k a CONTIGUOUS range of
// (frame, item) pairs instead -- whole frames -- so a line is fetched once and then found in that XCD's L2.  Speed only: the
// result does not depend on where a workgroup runs.  on == 0 keeps the natural order, 1 = one contiguous range per XCD,
// C > 1 = chunks of C consecutive items per XCD, dealt cyclically (CCM_ORB_XCD / _FC / _OD, for A/B timing).
__device__ __forceinline__ void xcd_work_item(int on, int& x, int& y)
{
    x = (int)blockIdx.x; y = (int)blockIdx.y;
    if (!on) return;
    const unsigned gx = gridDim.x, total = gx * gridDim.y;
    const unsigned lin = blockIdx.y * gx + blockIdx.x;
    unsigned w;
    if (on == 1) {                                                         // each XCD one contiguous eighth of the items
        const unsigned xcd = lin & 7u, q = total >> 3, rem = total & 7u;
        w = xcd * q + (xcd < rem ? xcd : rem) + (lin >> 3);                // start of the XCD's range + its (lin / 8)-th slot
    } else {                                                               // chunk-cyclic: chunks of `on` consecutive items go to one XCD, chunk c to XCD c % 8
        const unsigned C = (unsigned)on, span = 8u * C;
        const unsigned full = total - total % span;                        // the ragged tail keeps the natural order
        if (lin >= full) w = lin;
        else {
            const unsigned xcd = lin & 7u, j = lin >> 3;                   // the XCD's j-th workgroup overall
            const unsigned super = j / C, within = j - super * C;         // j-th = `within` of the XCD's chunk number `super`
            w = (super * 8u + xcd) * C + within;
        }
    }
    // w / gx without the ~40-instruction integer division in front of every workgroup's first load (w < 2^24: exact in float)
    unsigned q = (unsigned)((float)w * __frcp_rn((float)gx));
    if (q * gx > w) q--; else if ((q + 1u) * gx <= w) q++;
    y = (int)q; x = (int)(w - q * gx);
}

// inclusive wave prefix sum
__device__ __forceinline__ int wave_incl_scan(int v)
{
    const int lane = lane_id();
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        int y = __shfl_up(v, d, 64);
        if (lane >= d) v += y;
    }
    return v;
}
__device__ __forceinline__ int wave_sum(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d, 64);
    return v;
}

// ------------------------------------------------------------------------------------------------
// k_pyr_resize: one thread = 4 horizontally adjacent output pixels in each of RS_ROWS consecutive rows: the x tables
// are loaded once, and the 2 * RS_ROWS source loads are independent and issued together (the kernel is bound by
// dependent-load latency, not by bytes or ALU).
// Fixed-point bilinear exactly as cv::resize(INTER_LINEAR) for 8-bit data:
// weights are the host-made 11-bit tables, horizontal pass int32, vertical pass
// ((b0*(r0>>4))>>16) + ((b1*(r1>>4))>>16) + 2 >> 2.
// Round 3 built the streaming alternative -- a wave walks 32 output rows of its 256-column strip, loads every source row once (6 rows
// requested ahead), keeps its horizontal pass for the two output rows that need it (3.6 instead of 6 loaded dwords per output row) --
// bit-exact, and slower: 0.270 against 0.173 ms for the seven levels.  Its serial row loop needs selects, moves and address arithmetic
// that the fully unrolled kernel below does not (18 against 14 vector instructions per pixel), and a quarter of the waves.  Not kept.
#ifndef RS_ROWS
#define RS_ROWS 8                // rows per thread (measured: 4 -> 0.186, 8 -> 0.172, 16 -> 0.260 ms)
#endif
__global__ __launch_bounds__(256) void k_pyr_resize(const OrbGeom g, int level)
{
    const OrbLevel& D = g.lv[level];
    const OrbLevel& S = g.lv[level - 1];
    const int lane = threadIdx.x & 63;
    const int x4 = (blockIdx.x * 64 + lane) * 4;
    // everything that depends on the row only is wave-uniform: kept in scalar registers
    const int yb = (blockIdx.y * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6)) * RS_ROWS;
    const int f = blockIdx.z + g.frame0;
    if (yb >= D.h) return;
    const uint8_t* src = S.img + (long long)f * S.plane;
    uint8_t* dstp = const_cast<uint8_t*>(D.img) + (long long)f * D.plane;
    // lanes 0..RS_ROWS-1 fetch the y tables of the wave's rows; v_readlane broadcasts them
    int yo; unsigned ybb;
    {
        const int y = min(yb + (lane & (RS_ROWS - 1)), D.h - 1);
        yo = D.yofs[y];
        __builtin_memcpy(&ybb, D.yab + 2 * y, 4);
    }
    const uint8_t* r0p[RS_ROWS]; const uint8_t* r1p[RS_ROWS]; unsigned b0[RS_ROWS], b1[RS_ROWS];
#pragma unroll
    for (int r = 0; r < RS_ROWS; r++) {
        const int sy0 = __builtin_amdgcn_readlane(yo, r);
        const unsigned bb = (unsigned)__builtin_amdgcn_readlane((int)ybb, r);
        r0p[r] = src + (long long)sy0 * S.pitch;
        r1p[r] = src + (long long)min(sy0 + 1, S.h - 1) * S.pitch;
        b0[r] = bb & 0xFFFFu; b1[r] = bb >> 16;
    }
    if (x4 >= D.w) return;
    {
        // tables of the 4 outputs: two 16-byte loads (the tables are only 4-/2-byte aligned); the last, partial quad of
        // a row repeats its last column (the bytes past D.w land in the row padding, which nothing reads as data)
        int sxs[4]; unsigned abw[4];
        if (x4 + 3 < D.w) {
            int4 so; uint4 ab;
            __builtin_memcpy(&so, D.xofs + x4, 16);
            __builtin_memcpy(&ab, D.xab + 2 * x4, 16);
            sxs[0] = so.x; sxs[1] = so.y; sxs[2] = so.z; sxs[3] = so.w;
            abw[0] = ab.x; abw[1] = ab.y; abw[2] = ab.z; abw[3] = ab.w;
        } else {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const int x = min(x4 + i, D.w - 1);
                sxs[i] = D.xofs[x];
                __builtin_memcpy(&abw[i], D.xab + 2 * x, 4);
            }
        }
        // The 4 outputs read source columns base .. base+7 at most.  At the right border the 8-byte window is pulled
        // back inside the row (the neighbour weight of the last column is 0, so its clamped byte is never used), and
        // the 12-byte fetch is pulled back inside the pitch: every lane takes this path, no divergent border code.
        const int base = min(sxs[0], S.w - 8);
        const int ba = min(base & ~3, S.pitch - 12);
        if (S.w >= 8 && S.pitch >= 12 && sxs[3] - base <= 7) {
            uint2 q0[RS_ROWS], q1[RS_ROWS];
            const unsigned sh = (unsigned)(base - ba);            // 0..4
            const bool sh4 = sh > 3u;
#pragma unroll
            for (int r = 0; r < RS_ROWS; r++) {
                const unsigned* p0 = reinterpret_cast<const unsigned*>(r0p[r] + (unsigned)ba);
                const unsigned* p1 = reinterpret_cast<const unsigned*>(r1p[r] + (unsigned)ba);
                const unsigned d00 = p0[0], d01 = p0[1], d02 = p0[2], d10 = p1[0], d11 = p1[1], d12 = p1[2];
                const unsigned l0 = sh4 ? d01 : d00, m0 = sh4 ? d02 : d01, l1 = sh4 ? d11 : d10, m1 = sh4 ? d12 : d11;
                q0[r].x = __builtin_amdgcn_alignbyte(m0, l0, sh); q0[r].y = __builtin_amdgcn_alignbyte(d02, m0, sh);   // v_alignbyte uses sh & 3
                q1[r].x = __builtin_amdgcn_alignbyte(m1, l1, sh); q1[r].y = __builtin_amdgcn_alignbyte(d12, m1, sh);
            }
            unsigned sel[4];
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const unsigned o = (unsigned)(sxs[i] - base);
                sel[i] = 0x0c000c00u + o + (min(o + 1u, 7u) << 16);   // v_perm_b32: bytes o and o+1 as a u16 pair
            }
#pragma unroll
            for (int r = 0; r < RS_ROWS; r++) {
                unsigned out = 0;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    // v_dot2_u32_u16 applies the packed (a0,a1) weights
                    const unsigned h0 = __builtin_amdgcn_udot2(fc_pk(__builtin_amdgcn_perm(q0[r].y, q0[r].x, sel[i])), fc_pk(abw[i]), 0u, false);
                    const unsigned h1 = __builtin_amdgcn_udot2(fc_pk(__builtin_amdgcn_perm(q1[r].y, q1[r].x, sel[i])), fc_pk(abw[i]), 0u, false);
                    // b <= 2048 and h >> 4 < 2^15: 24-bit multiplies are exact; the result is <= 255
                    const unsigned v = ((__umul24(b0[r], h0 >> 4) >> 16) + (__umul24(b1[r], h1 >> 4) >> 16) + 2u) >> 2;
                    out |= v << (8 * i);
                }
                if (yb + r < D.h) *reinterpret_cast<unsigned*>(dstp + (long long)(yb + r) * D.pitch + (unsigned)x4) = out;
            }
            return;
        }
    }
    for (int r = 0; r < RS_ROWS && yb + r < D.h; r++) {
        unsigned out = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int x = min(x4 + i, D.w - 1);
            const int sx0 = D.xofs[x];
            const int sx1 = min(sx0 + 1, S.w - 1);
            const int a0 = D.xab[2 * x], a1 = D.xab[2 * x + 1];
            const int h0 = r0p[r][sx0] * a0 + r0p[r][sx1] * a1;
            const int h1 = r1p[r][sx0] * a0 + r1p[r][sx1] * a1;
            const int v = ((((int)b0[r] * (h0 >> 4)) >> 16) + (((int)b1[r] * (h1 >> 4)) >> 16) + 2) >> 2;
            out |= (unsigned)(v & 255) << (8 * i);
        }
        // our level buffers have pitch % 64 == 0 and a 256-byte aligned base: the dword store is aligned
        *reinterpret_cast<unsigned*>(dstp + (long long)(yb + r) * D.pitch + (unsigned)x4) = out;
    }
}

// ------------------------------------------------------------------------------------------------
// FAST-9/16 score: the largest threshold t for which the pixel is still a corner,
// i.e. max over the 16 arcs of 9 contiguous ring pixels of min(v - ring) (ring darker) or of
// min(ring - v) (ring brighter), minus one.  A pixel is a corner at threshold T iff score >= T, so
// one score map serves iniThFAST and the minThFAST fallback.  Equivalent to cv::FAST's
// cornerScore<16> for every detected corner (checked against the oracle's literal restatement).
__host__ __device__ inline int fast_score16(int v, const int* r)
{
    int d[16];
#pragma unroll
    for (int k = 0; k < 16; k++) d[k] = v - r[k];
    int lo3[16], hi3[16];
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int a = d[k], b = d[(k + 1) & 15], c = d[(k + 2) & 15];
        lo3[k] = min(a, min(b, c));
        hi3[k] = max(a, max(b, c));
    }
    int A = -256, B = 256;
#pragma unroll
    for (int k = 0; k < 16; k++) {
        A = max(A, min(lo3[k], min(lo3[(k + 3) & 15], lo3[(k + 6) & 15])));
        B = min(B, max(hi3[k], max(hi3[(k + 3) & 15], hi3[(k + 6) & 15])));
    }
    return max(A, -B) - 1;
}

#define ST_W 64
#define ST_H 64            // tile height: ST_H/16 pixel rows per thread
#define ST_LW 72           // 64 + 2*4 (3-px halo rounded up to a dword on each side)
#define ST_LH (ST_H + 6)

// Phase 1: every thread applies the cheap rejection test to its pixels (4 adjacent pixels in each of
// ST_H/16 rows) and appends survivors to an LDS list (wave ballot + one LDS atomic per wave).  Phase 2:
// the list is processed densely, one survivor per lane, so the 100-instruction score never runs on a
// mostly idle wave.  Phase 3: the score tile is written as dwords.
__global__ __launch_bounds__(256) void k_fast_score(const OrbGeom g)
{
    __shared__ __attribute__((aligned(16))) uint8_t tile[ST_LH][ST_LW];
    __shared__ __attribute__((aligned(16))) uint8_t outt[ST_H][ST_W];
    __shared__ unsigned short surv[ST_W * ST_H];
    __shared__ int nsurv;
    // flattened tile index -> level
    int t = blockIdx.x, level = 0;
    for (int l = 1; l < g.nlevels; l++)
        if (t >= g.lv[l].tile_first) level = l;
    const OrbLevel& L = g.lv[level];
    t -= L.tile_first;
    const int tx0 = (t % L.tiles_x) * ST_W, ty0 = (t / L.tiles_x) * ST_H;
    const int f = blockIdx.y + g.frame0;
    const uint8_t* img = L.img + (long long)f * L.plane;
    if (threadIdx.x == 0) nsurv = 0;
    for (int i = threadIdx.x; i < ST_H * ST_W / 4; i += 256) reinterpret_cast<unsigned*>(&outt[0][0])[i] = 0u;
    // stage (clamped) pixels; clamped duplicates are only read by pixels whose score is not needed
    const bool dword_ok = ((reinterpret_cast<uintptr_t>(img) | (uintptr_t)L.pitch) & 3) == 0 && L.pitch >= ((L.w + 3) & ~3);
    if (dword_ok) {
        const int wmax = ((L.w - 1) & ~3);
        for (int i = threadIdx.x; i < ST_LH * (ST_LW / 4); i += 256) {
            const int ly = i / (ST_LW / 4), lx = i - ly * (ST_LW / 4);
            const int gy = min(max(ty0 + ly - 3, 0), L.h - 1);
            const int gx = min(max(tx0 + 4 * lx - 4, 0), wmax);
            reinterpret_cast<unsigned*>(&tile[ly][0])[lx] = *reinterpret_cast<const unsigned*>(img + (long long)gy * L.pitch + gx);
        }
    } else {
        for (int i = threadIdx.x; i < ST_LH * ST_LW; i += 256) {
            const int ly = i / ST_LW, lx = i - ly * ST_LW;
            const int gy = min(max(ty0 + ly - 3, 0), L.h - 1);
            const int gx = min(max(tx0 + lx - 4, 0), L.w - 1);
            tile[ly][lx] = img[(long long)gy * L.pitch + gx];
        }
    }
    __syncthreads();
    const int px = (threadIdx.x & 15) * 4;
    const int t_lo = min(g.ini_th, g.min_th);
    for (int py = threadIdx.x >> 4; py < ST_H; py += 16) {
        const int gy = ty0 + py;
        const bool row_ok = gy >= ORB_EDGE && gy < L.h - ORB_EDGE;
        const int cy = py + 3;
        // the 12 bytes around the 4 centre pixels and the 4 bytes three rows below / above, as dwords
        const unsigned* crow = reinterpret_cast<const unsigned*>(&tile[cy][0]) + (px >> 2);
        const unsigned c0 = crow[0], c1 = crow[1], c2 = crow[2];
        const unsigned nn = reinterpret_cast<const unsigned*>(&tile[cy + 3][0])[(px >> 2) + 1];
        const unsigned ss = reinterpret_cast<const unsigned*>(&tile[cy - 3][0])[(px >> 2) + 1];
        const unsigned long long lo = ((unsigned long long)c1 << 32) | c0, hi = ((unsigned long long)c2 << 32) | c1;
        unsigned keep4 = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int gx = tx0 + px + i;
            if (row_ok && gx >= ORB_EDGE && gx < L.w - ORB_EDGE) {
                const int v = (int)((c1 >> (8 * i)) & 0xFF);
                const int n = (int)((nn >> (8 * i)) & 0xFF), s = (int)((ss >> (8 * i)) & 0xFF);
                const int w = (int)((lo >> (8 * (i + 1))) & 0xFF);         // column cx-3 = byte 4+i-3 of the 12
                const int e = (int)((hi >> (8 * (i + 3))) & 0xFF);         // column cx+3 = byte 4+i+3
                // every 9-arc contains ring pixel k or k+8: both within +-t_lo -> never a corner
                const bool keep = !((abs(v - n) <= t_lo && abs(v - s) <= t_lo) || (abs(v - e) <= t_lo && abs(v - w) <= t_lo));
                keep4 |= keep ? (1u << i) : 0u;
            }
        }
        // survivors are rare: one ballot decides for the whole wave-row
        if (__ballot(keep4 != 0u) != 0ull) {
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const bool keep = (keep4 >> i) & 1u;
                const unsigned long long m = __ballot(keep);
                if (m != 0ull) {
                    int base = 0;
                    if (lane_id() == 0) base = atomicAdd(&nsurv, __popcll(m));
                    base = __shfl(base, 0, 64);
                    if (keep) surv[base + __popcll(m & lanemask_lt())] = (unsigned short)(py * ST_W + px + i);
                }
            }
        }
    }
    __syncthreads();
    const int ns = nsurv;
    for (int si = threadIdx.x; si < ns; si += 256) {
        const int pos = surv[si];
        const int cy = (pos >> 6) + 3, cx = (pos & 63) + 4;
        const int v = tile[cy][cx];
        int r[16];
        r[0] = tile[cy + 3][cx];      r[1] = tile[cy + 3][cx + 1];  r[2] = tile[cy + 2][cx + 2];
        r[3] = tile[cy + 1][cx + 3];  r[4] = tile[cy][cx + 3];      r[5] = tile[cy - 1][cx + 3];
        r[6] = tile[cy - 2][cx + 2];  r[7] = tile[cy - 3][cx + 1];  r[8] = tile[cy - 3][cx];
        r[9] = tile[cy - 3][cx - 1];  r[10] = tile[cy - 2][cx - 2]; r[11] = tile[cy - 1][cx - 3];
        r[12] = tile[cy][cx - 3];     r[13] = tile[cy + 1][cx - 3]; r[14] = tile[cy + 2][cx - 2];
        r[15] = tile[cy + 3][cx - 1];
        const int sc = fast_score16(v, r);
        if (sc >= t_lo && sc > 0) outt[pos >> 6][pos & 63] = (uint8_t)sc;
    }
    __syncthreads();
    for (int py = threadIdx.x >> 4; py < ST_H; py += 16) {
        const int gy = ty0 + py;
        if (gy < L.h && tx0 + px < L.spitch)
            *reinterpret_cast<unsigned*>(L.smap + (long long)f * L.splane + (long long)gy * L.spitch + tx0 + px) =
                reinterpret_cast<const unsigned*>(&outt[py][0])[px >> 2];
    }
}

// ------------------------------------------------------------------------------------------------
// k_cell_nms: one wave per FAST cell.  Reproduces, on the cell's sub-image, what
//   FAST(sub, iniThFAST, nms=true); if empty FAST(sub, minThFAST, nms=true)
// returns: detection area = sub-image minus a 3-px margin; scores of non-corners and of pixels
// outside the detection area count as 0; keep a corner iff its score is strictly greater than its 8
// neighbours'; output in row-major order.  Candidates go to the cell's slot range as
// score<<24 | y<<12 | x with x,y relative to minBorderX/Y (the coordinates of vToDistributeKeys).
#define NMS_PITCH 72      // interior starts at byte 4 of a row (dword aligned), zero ring at byte 3 and after the last column
#define NMS_ROWS 62       // cells are at most 65 px: 59 detection rows + 2 ring rows
#define NMS_WAVE_LDS (NMS_PITCH * NMS_ROWS)
__global__ __launch_bounds__(256) void k_cell_nms(const OrbGeom g, const OrbCell* __restrict__ cells,
                                                  unsigned* __restrict__ slots, int* __restrict__ cell_count)
{
    __shared__ __attribute__((aligned(16))) uint8_t lds[4][NMS_WAVE_LDS];
    const int wv = threadIdx.x >> 6, lane = lane_id();
    const int ci = blockIdx.x * 4 + wv;
    if (ci >= g.ncells) return;
    const int f = blockIdx.y + g.frame0;
    const OrbCell c = cells[ci];
    const OrbLevel& L = g.lv[c.level];
    const int rx0 = c.x0 + 3, ry0 = c.y0 + 3, rw = c.cw - 6, rh = c.ch - 6;
    int* count_out = cell_count + (long long)f * g.ncells + ci;
    if (rw <= 0 || rh <= 0) { if (lane == 0) *count_out = 0; return; }
    uint8_t* T = lds[wv];
    const uint8_t* sm = L.smap + (long long)f * L.splane;
    // rows 0 and rh+1 of the tile are the zero ring
    for (int i = lane; i < NMS_PITCH / 4; i += 64) {
        reinterpret_cast<unsigned*>(T)[i] = 0u;
        reinterpret_cast<unsigned*>(T + (rh + 1) * NMS_PITCH)[i] = 0u;
    }
    // interior: 16 lanes x 4 bytes per row (rw <= 59), 4 rows per step; bytes beyond rw are zeroed (they belong
    // to the neighbouring cell).  The score map has slack behind every plane, so the 4-byte over-read is safe.
    const int lx = (lane & 15) * 4, lrow = lane >> 4;
    bool any = false;
    {
        // all loads are issued before the first LDS store so that their latencies overlap (rh <= 59: 15 steps)
        unsigned v[15];
#pragma unroll
        for (int it = 0; it < 15; it++) {
            const int yy = 4 * it + lrow;
            v[it] = 0;
            if (yy < rh && lx < rw) __builtin_memcpy(&v[it], sm + (long long)(ry0 + yy) * L.spitch + rx0 + lx, 4);
        }
        const int valid = rw - lx;
        const unsigned keep_mask = valid >= 4 ? 0xFFFFFFFFu : (valid <= 0 ? 0u : (1u << (8 * valid)) - 1u);
#pragma unroll
        for (int it = 0; it < 15; it++) {
            const int yy = 4 * it + lrow;
            if (yy < rh) {
                const unsigned w = v[it] & keep_mask;
                *reinterpret_cast<unsigned*>(T + (yy + 1) * NMS_PITCH + 4 + lx) = w;
                if ((lane & 15) == 0) *reinterpret_cast<unsigned*>(T + (yy + 1) * NMS_PITCH) = 0u;   // left ring
                any |= w != 0u;
            }
        }
    }
    if (__ballot(any) == 0ull) { if (lane == 0) *count_out = 0; return; }      // no score >= min(ini,min) anywhere
    __builtin_amdgcn_s_waitcnt(0xc07f);     // lgkmcnt(0): this wave's LDS stores have landed
    __builtin_amdgcn_wave_barrier();
    unsigned* out = slots + (long long)f * g.slots_per_frame + c.slot_first;
    const int relx = rx0 - ORB_BORDER, rely = ry0 - ORB_BORDER;
    int n = 0;
    // two rows per step when a row fits 32 lanes; ballot bit order == row-major scan order
    const int rpi = rw <= 32 ? 2 : 1;
    const int xx = rpi == 2 ? (lane & 31) : lane, yoff = rpi == 2 ? (lane >> 5) : 0;
    // First iniThFAST; the fallback to minThFAST happens when FAST returned NO KEYPOINT, i.e. after
    // non-max suppression (:981) -- corners that suppress each other with equal scores also trigger it.
    for (int attempt = 0; attempt < 2 && n == 0; attempt++) {
        const int th = attempt == 0 ? g.ini_th : g.min_th;
        for (int y0 = 0; y0 < rh; y0 += rpi) {
            const int yy = y0 + yoff;
            bool keep = false;
            int s = 0;
            const uint8_t* p = &T[(min(yy, rh - 1) + 1) * NMS_PITCH + 4 + xx];
            if (yy < rh && xx < rw) s = p[0];
            // most rows hold no corner at all: skip the 8 neighbour reads for the whole wave
            if (__ballot(s >= th && s > 0) == 0ull) continue;
            {
                if (s >= th && s > 0) {
                    // scores below the threshold belong to non-corners and count as 0
#define NB(o) ((int)p[o] >= th ? (int)p[o] : 0)
                    keep = s > NB(-1) && s > NB(1) &&
                           s > NB(-NMS_PITCH - 1) && s > NB(-NMS_PITCH) && s > NB(-NMS_PITCH + 1) &&
                           s > NB(NMS_PITCH - 1) && s > NB(NMS_PITCH) && s > NB(NMS_PITCH + 1);
#undef NB
                }
            }
            const unsigned long long m = __ballot(keep);
            if (keep) {
                const int pos = n + __popcll(m & lanemask_lt());
                if (pos < c.slot_cap)
                    out[pos] = ((unsigned)s << 24) | ((unsigned)(rely + yy) << 12) | (unsigned)(relx + xx);
            }
            n += __popcll(m);
        }
    }
    if (lane == 0) *count_out = n;
}

// ------------------------------------------------------------------------------------------------
// k_fast_cells: k_fast_score + k_cell_nms fused on a band = a run of cells of one cell row.  The band's pixels
// are staged once in LDS, scored in LDS (rejection test -> survivor list -> dense scoring, in row blocks), and
// each cell's threshold choice / 3x3 NMS / ordered compaction runs straight from the LDS score tile: the score
// map never goes to HBM.  Bands of adjacent cell rows overlap by 6 rows, which are scored twice.
// One band per workgroup.  Round 3 built the alternative -- a workgroup loops over 1..8 consecutive bands and requests the next
// band's pixels (8 registers per thread, one branch-free 16-byte global load per chunk) as soon as the current band is scored, so
// that they arrive under the NMS -- and measured it on one box against this kernel (tools/ab_orb.sh): 0.391 ms at 1 band per
// workgroup, 0.410 / 0.421 / 0.424 / 0.436 / 0.448 at 2 / 3 / 4 / 6 / 8, against 0.368.  A round of the loop (the extra barrier, the
// band record and the kernel arguments re-read through scalar loads, 64 registers instead of 54) costs more than a fresh workgroup,
// whose start-up the dispatcher overlaps with the seven others on the CU.  Not kept.
// NMS works on the list of scored pixels (FC_NZ entries) and the list of local maxima (FC_KEPT); a band that overflows
// either list takes the per-cell row scan instead.
#ifndef FC_NZ
#define FC_NZ 1024
#endif
#ifndef FC_KEPT
#define FC_KEPT 480             // + the per-cell bucket counters = the 20480 bytes of LDS that let 8 workgroups share a CU on the 4-cell bands
#endif
#define FC_CELLS 32
#ifndef FC_TPB
#define FC_TPB 256              // threads per band workgroup (measured: 192 0.56, 256 0.52, 320 0.75, 384 0.82 ms)
#endif
// Survivors of the rejection test wait on the wave's own stack in LDS (running count in a scalar register, no atomic) and are scored
// 64 at a time: the stack never holds more than 63 waiting entries + the 256 pixels of one run of 64 four-pixel items.
// (surv_cap: rounds 1-2 pooled the survivors of a row block over the workgroup; the parameter is kept for the launcher's signature.)
__host__ __device__ inline int fc_wave_cap(int surv_cap) { (void)surv_cap; return 320; }      // wave-local stack: at most 63 waiting + the 256 pixels of one item
__host__ __device__ inline size_t fc_lds_bytes(int pitch, int bh, int surv_cap)
{
    return 2 * (size_t)pitch * bh + (size_t)fc_wave_cap(surv_cap) * (FC_TPB / 64) * 2 + 32 + FC_NZ * 2 + FC_KEPT * 4 + FC_CELLS * 16 + 64 * 8 + 256 + 64;
}


// PACKED: the rejection test on two pixels per register with the packed 16-bit VALU (v_pk_sub_u16 / v_pk_max_u16 /
// v_pk_min_u16): |v - n| <= t  <=>  (u16)(v + t - n) <= 2t.
__device__ __forceinline__ int fc_mbcnt(unsigned long long m, int base)
{
    return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, (unsigned)base));
}
// Diagnostic build only (-DFC_STAMPS, tools/r03_fc_stamps.sh): wave 0 of every workgroup leaves the shader clock at its phase
// boundaries in a buffer no other code reads; the shipped kernel contains none of this.
#ifdef FC_STAMPS
#define FC_STAMP_SLOTS 8
#define FC_STAMP_WGS (1 << 17)
__device__ unsigned long long fc_stamps[FC_STAMP_WGS * FC_STAMP_SLOTS];
#define FC_STAMP(k) do { if (threadIdx.x == 0) { const unsigned wg__ = blockIdx.y * gridDim.x + blockIdx.x; if (wg__ < FC_STAMP_WGS) fc_stamps[wg__ * FC_STAMP_SLOTS + (k)] = __builtin_amdgcn_s_memtime(); } } while (0)
extern "C" int ccm_debug_fc_stamps(unsigned long long* out, int n_wgs)
{
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(fc_stamps), (size_t)n_wgs * FC_STAMP_SLOTS * 8, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : -1;
}
#else
#define FC_STAMP(k) do { } while (0)
#endif
// element ci (per lane) of four wave-uniform values: three selects, no memory
template <class T>
__device__ __forceinline__ int sel4(int ci, const T (&a)[4]) { return ci == 0 ? (int)a[0] : ci == 1 ? (int)a[1] : ci == 2 ? (int)a[2] : (int)a[3]; }
template <bool PACKED>
__global__ __launch_bounds__(FC_TPB) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_fast_cells(const OrbGeom g, const OrbCell* __restrict__ cells, const OrbBand* __restrict__ bands,
                                                    unsigned* __restrict__ slots, int* __restrict__ cell_count, int unused_, int abl, int xcd_on)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t fc_smem[];
    int wx, wy;
    xcd_work_item(xcd_on, wx, wy);
    FC_STAMP(0);
    const OrbBand B = bands[wx];
#ifdef FC_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (diagnostic: the band descriptor has arrived)
#endif
    FC_STAMP(6);
    // the band's cells, held in scalar registers from here on (left to itself the compiler loads them again where the NMS uses them:
    // another scalar-load latency in each of its two passes, +1,400 cycles per band measured)
    int bclo[4], bcwd[4], bsfirst[4], bscap[4];
#pragma unroll
    for (int i = 0; i < 4; i++) {
        bclo[i] = B.clo[i]; bcwd[i] = B.cwd[i]; bsfirst[i] = B.slot_first[i]; bscap[i] = B.slot_cap[i];
        asm volatile("" : "+s"(bclo[i]), "+s"(bcwd[i]), "+s"(bsfirst[i]), "+s"(bscap[i]));
    }
    int bxa = B.xa, by0 = B.y0, bncells = B.ncells, bcfirst = B.cell_first;
    asm volatile("" : "+s"(bxa), "+s"(by0), "+s"(bncells), "+s"(bcfirst));
    const int f = wy + g.frame0, tid = threadIdx.x, lane = lane_id(), wv = tid >> 6;
    // the level's image: from the band record, except level 0 (the caller's frames: pointer, pitch and plane are per call and sit at a
    // fixed place of the kernel arguments -- no load that depends on the band record)
    const bool lvl0 = B.level == 0;
    const uint8_t* const Limg = lvl0 ? g.lv[0].img : B.img;
    const long long Lplane = lvl0 ? g.lv[0].plane : B.plane;
    const int Lpitch = lvl0 ? g.lv[0].pitch : B.lpitch, Lw = B.w, Lh = B.h;
    const int P = B.pitch, bh = B.bh, PW = P >> 2;
    uint8_t* T = fc_smem;                                   // pixels  [bh][P]
    uint8_t* S = fc_smem + (size_t)P * bh;                  // scores  [bh][P]
    constexpr int NW = FC_TPB / 64;
    const int WCAP = fc_wave_cap(0);
    unsigned short* surv = reinterpret_cast<unsigned short*>(S + (size_t)P * bh);     // [NW][WCAP]
    int* nsurv = reinterpret_cast<int*>(surv + NW * WCAP);   // [1] scored pixels, [2] local maxima, [4 + w] survivors of wave w in the row block
    unsigned short* nz = reinterpret_cast<unsigned short*>(nsurv + 8);
    unsigned* kept = reinterpret_cast<unsigned*>(nz + FC_NZ);
    int* cell_hi = reinterpret_cast<int*>(kept + FC_KEPT);   // per cell: local maxima with score >= iniThFAST
    int* cell_n = cell_hi + FC_CELLS;                        // per cell: keypoints written
    int* cell_k = cell_n + FC_CELLS;                         // per cell: local maxima in the cell's bucket of `kept`
    uint2* colmask = reinterpret_cast<uint2*>(cell_k + FC_CELLS + FC_CELLS);     // per tile dword: 16-bit lane masks of its detection columns: .x pixels 0 and 2, .y pixels 1 and 3
    // (what the NMS needs of a cell -- its detection columns and its candidate slots -- comes with the band record: until round 3 the
    //  cell records were loaded from global memory, first on the NMS's critical path, then in front of the pixel loads)
    uint8_t* colcell = reinterpret_cast<uint8_t*>(colmask + 64);     // [256] cell of a tile column; 255 = no cell's detection column
    if (tid < FC_CELLS) { cell_hi[tid] = 0; cell_n[tid] = 0; cell_k[tid] = 0; }
    // per cell, for the NMS (indexed per lane there: an LDS read measured 0.7 % faster than three selects on the scalar registers)
    int* l_clo = reinterpret_cast<int*>(colcell + 256); int* l_cwd = l_clo + 4; int* l_sfirst = l_cwd + 4; int* l_scap = l_sfirst + 4;
    if (tid < 4) { l_clo[tid] = sel4(tid, bclo); l_cwd[tid] = sel4(tid, bcwd); l_sfirst[tid] = sel4(tid, bsfirst); l_scap[tid] = sel4(tid, bscap); }
    if (tid == 0) { nsurv[1] = 0; nsurv[2] = 0; }
    const uint8_t* img = Limg + (long long)f * Lplane;
    // ---- stage pixels (clamped: duplicates are only read for pixels whose score is not needed), zero the scores
    const bool dword_ok = ((reinterpret_cast<uintptr_t>(img) | (uintptr_t)Lpitch) & 3) == 0 && Lpitch >= ((Lw + 3) & ~3);
    // i / PW by multiply-shift: exact for i < 2^20 / PW (i <= 65 * 64 here)
    const unsigned pw_inv = B.pw_inv;
#ifdef FC_STAMPS
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (diagnostic: level record and everything scalar before the pixel loads has arrived)
#endif
    FC_STAMP(5);
    if (dword_ok && (P & 15) == 0) {
        // 16 bytes per lane: one global load and two ds_write_b128 (pixels, zeroed scores) per 16 pixels; the last
        // chunks of a row are clamped dword by dword
        const int wmax = ((Lw - 1) & ~3);
        const int PQ = P >> 4;
        const unsigned pq_inv = B.pq_inv;                          // i / PQ exact for i < 2^20 / PQ
        for (int i = tid; i < bh * PQ; i += FC_TPB) {
            const int ly = (int)(((unsigned)i * pq_inv) >> 20), lq = i - ly * PQ;
            const int gy = min(B.y0 + ly, Lh - 1), gx = B.xa + 16 * lq;
            const uint8_t* rowp = img + (long long)gy * Lpitch;
            uint4 v;
            if (gx + 12 <= wmax) __builtin_memcpy(&v, rowp + gx, 16);
            else {
                v.x = *reinterpret_cast<const unsigned*>(rowp + min(gx, wmax));
                v.y = *reinterpret_cast<const unsigned*>(rowp + min(gx + 4, wmax));
                v.z = *reinterpret_cast<const unsigned*>(rowp + min(gx + 8, wmax));
                v.w = *reinterpret_cast<const unsigned*>(rowp + min(gx + 12, wmax));
            }
            reinterpret_cast<uint4*>(T)[i] = v;
            reinterpret_cast<uint4*>(S)[i] = make_uint4(0u, 0u, 0u, 0u);
        }
    } else if (dword_ok) {
        const int wmax = ((Lw - 1) & ~3);
        for (int i = tid; i < bh * PW; i += FC_TPB) {
            const int ly = (int)(((unsigned)i * pw_inv) >> 20), lx = i - ly * PW;
            const int gy = min(B.y0 + ly, Lh - 1), gx = min(B.xa + 4 * lx, wmax);
            reinterpret_cast<unsigned*>(T)[i] = *reinterpret_cast<const unsigned*>(img + (long long)gy * Lpitch + gx);
            reinterpret_cast<unsigned*>(S)[i] = 0u;
        }
    } else {
        for (int i = tid; i < bh * P; i += FC_TPB) {
            const int ly = i / P, lx = i - ly * P;
            const int gy = min(B.y0 + ly, Lh - 1), gx = min(B.xa + lx, Lw - 1);
            T[i] = img[(long long)gy * Lpitch + gx];
            S[i] = 0;
        }
    }
#ifdef FC_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // (diagnostic: this wave's pixels have arrived)
#endif
    FC_STAMP(7);
    // columns of the tile that belong to some cell's detection area: [c_lo, c_hi)
    const int c_lo = B.c_lo, c_hi = B.c_hi;
    const int t_lo = min(g.ini_th, g.min_th);
    // dwords that hold at least one detection column and have both neighbours inside the row
    const int dw_lo = B.dw_lo, dw_hi = B.dw_hi;
    if (tid < PW) {
        unsigned mk[4];
        for (int i = 0; i < 4; i++) mk[i] = (4 * tid + i >= c_lo && 4 * tid + i < c_hi) ? 0xFFFFu : 0u;
        colmask[tid] = make_uint2(mk[0] | (mk[2] << 16), mk[1] | (mk[3] << 16));
    }
    __syncthreads();
    FC_STAMP(1);
    if (abl == 1) return;                                    // staging only
    if (tid < P) {                                           // (read after the next barrier)
        int ci = 255;
#pragma unroll
        for (int i = 0; i < ORB_BAND_CELLS; i++) if (i < bncells && tid >= bclo[i] && tid < bclo[i] + bcwd[i]) ci = i;
        colcell[tid] = (uint8_t)ci;
    }
    unsigned short* wsurv = surv + wv * WCAP;
    // ---- rejection + scoring, WAVE-LOCAL (round 3): a wave tests its own runs of 64 items (4 pixels per lane), pushes the survivors on
    // its own stack in LDS, and scores them 64 at a time -- a full wave per 130-instruction score -- as soon as 64 are waiting.  No
    // workgroup barrier until every pixel of the band is scored: round 2 pooled the survivors of a row block over the four waves
    // (two barriers per row block, and every wave waited for the slowest at each of them).  The stack never holds more than
    // 63 + 256 entries.  The order in which pixels are scored does not matter: the NMS below works on the set.
    auto score_at = [&](int pos) {
        const uint8_t* c = T + pos;
        const int v = c[0];
        int r[16];
        r[0] = c[3 * P];      r[1] = c[3 * P + 1];  r[2] = c[2 * P + 2];  r[3] = c[P + 3];
        r[4] = c[3];          r[5] = c[-P + 3];     r[6] = c[-2 * P + 2]; r[7] = c[-3 * P + 1];
        r[8] = c[-3 * P];     r[9] = c[-3 * P - 1]; r[10] = c[-2 * P - 2]; r[11] = c[-P - 3];
        r[12] = c[-3];        r[13] = c[P - 3];     r[14] = c[2 * P - 2]; r[15] = c[3 * P - 1];
        const int sc = fast_score16(v, r);
        if (sc >= t_lo && sc > 0) {
            S[pos] = (uint8_t)sc;
            const int q = atomicAdd(nsurv + 1, 1);
            if (q < FC_NZ) nz[q] = (unsigned short)pos;
        }
    };
    {
        int wcnt = 0;                                          // entries on this wave's stack (wave-uniform)
        const int items = (bh - 6) * PW;
        constexpr int r0 = 3;
        for (int it0 = wv * 64; it0 < items; it0 += FC_TPB) {
            const int it = it0 + lane;
            const int rr = (int)(((unsigned)it * pw_inv) >> 20);
            const int row = r0 + rr, dw = it - rr * PW;
            const int px = 4 * dw;
            unsigned keep4 = 0;                                 // scalar path: bit i = pixel i survives
            unsigned kk0 = 0, kk1 = 0;                          // packed path: 16-bit halves, .lo/.hi of kk0 = pixels 0 / 2, of kk1 = pixels 1 / 3
            if (!PACKED && it < items && px + 3 >= c_lo && px < c_hi && dw >= 1 && dw + 1 < PW) {
                const unsigned* crow = reinterpret_cast<const unsigned*>(T + row * P) + dw;
                const unsigned c0 = crow[-1], c1 = crow[0], c2 = crow[1];
                const unsigned nn = reinterpret_cast<const unsigned*>(T + (row + 3) * P)[dw];
                const unsigned ss = reinterpret_cast<const unsigned*>(T + (row - 3) * P)[dw];
                const unsigned long long lo = ((unsigned long long)c1 << 32) | c0, hi = ((unsigned long long)c2 << 32) | c1;
#pragma unroll
                for (int i = 0; i < 4; i++) {
                    const int col = px + i;
                    if (col < c_lo || col >= c_hi) continue;
                    const int v = (int)((c1 >> (8 * i)) & 0xFF);
                    const int n = (int)((nn >> (8 * i)) & 0xFF), s2 = (int)((ss >> (8 * i)) & 0xFF);
                    const int w = (int)((lo >> (8 * (i + 1))) & 0xFF);         // column col-3 = byte 4+i-3 of c0|c1|c2
                    const int e = (int)((hi >> (8 * (i + 3))) & 0xFF);         // column col+3 = byte 4+i+3
                    const bool keep = !((abs(v - n) <= t_lo && abs(v - s2) <= t_lo) || (abs(v - e) <= t_lo && abs(v - w) <= t_lo));
                    keep4 |= keep ? (1u << i) : 0u;
                }
            }
            if (PACKED && it < items && dw >= dw_lo && dw <= dw_hi) {
                // P == 4 * PW: the centre dword of item `it` sits at byte r0 * P + 4 * it
                const unsigned* crow = reinterpret_cast<const unsigned*>(T + r0 * P) + it;
                const unsigned c0 = crow[-1], c1 = crow[0], c2 = crow[1];
                const unsigned nn = crow[3 * PW], ss = crow[-3 * PW];
                const unsigned ww = __builtin_amdgcn_alignbyte(c1, c0, 1);        // columns px-3 .. px
                const unsigned ee = __builtin_amdgcn_alignbyte(c2, c1, 3);        // columns px+3 .. px+6
                const fc_us2 tt = fc_pk((unsigned)t_lo * 0x00010001u);
                unsigned kk[2];
#pragma unroll
                for (int h = 0; h < 2; h++) {                                     // h = 0: pixels 0 and 2; h = 1: pixels 1 and 3
                    const unsigned sel = h ? 0x0c030c01u : 0x0c020c00u;           // v_perm_b32: two bytes -> two u16
                    const fc_us2 v = fc_pk(__builtin_amdgcn_perm(0u, c1, sel));
                    const fc_us2 n = fc_pk(__builtin_amdgcn_perm(0u, nn, sel)), sq = fc_pk(__builtin_amdgcn_perm(0u, ss, sel));
                    const fc_us2 w = fc_pk(__builtin_amdgcn_perm(0u, ww, sel)), e = fc_pk(__builtin_amdgcn_perm(0u, ee, sel));
                    // every 9-arc holds ring pixel k or k+8: a corner brighter (darker) than v by more than t needs the
                    // larger (smaller) of BOTH opposite pairs beyond v + t (v - t)
                    const fc_us2 bm = __builtin_elementwise_min(__builtin_elementwise_max(n, sq), __builtin_elementwise_max(e, w));
                    const fc_us2 dm = __builtin_elementwise_max(__builtin_elementwise_min(n, sq), __builtin_elementwise_min(e, w));
                    kk[h] = fc_u(__builtin_elementwise_sub_sat(bm, (fc_us2)(v + tt))) | fc_u(__builtin_elementwise_sub_sat(__builtin_elementwise_sub_sat(v, tt), dm));
                }
                const uint2 cm = colmask[dw];
                kk0 = kk[0] & cm.x; kk1 = kk[1] & cm.y;
            }
            // survivor predicates of the lane's four pixels: one 16-bit compare each on the packed path
            const bool k0 = PACKED ? (kk0 & 0xFFFFu) != 0u : (keep4 & 1u) != 0u, k1 = PACKED ? (kk1 & 0xFFFFu) != 0u : (keep4 & 2u) != 0u;
            const bool k2 = PACKED ? (kk0 >> 16) != 0u : (keep4 & 4u) != 0u, k3 = PACKED ? (kk1 >> 16) != 0u : (keep4 & 8u) != 0u;
            const unsigned long long m0 = __ballot(k0), m1 = __ballot(k1), m2 = __ballot(k2), m3 = __ballot(k3);
            if ((m0 | m1 | m2 | m3) != 0ull) {
                // stack slot of the lane's pixel i among the wave's survivors of this item: pixels 0 of all lanes first, then
                // pixels 1, ...; v_mbcnt adds the count of lower lanes to the running base
                const int b1 = wcnt + __popcll(m0), b2 = b1 + __popcll(m1), b3 = b2 + __popcll(m2);
                const int pos0 = r0 * P + 4 * it;                  // == row * P + px, as P == 4 * PW
                if (k0) wsurv[fc_mbcnt(m0, wcnt)] = (unsigned short)pos0;
                if (k1) wsurv[fc_mbcnt(m1, b1)] = (unsigned short)(pos0 + 1);
                if (k2) wsurv[fc_mbcnt(m2, b2)] = (unsigned short)(pos0 + 2);
                if (k3) wsurv[fc_mbcnt(m3, b3)] = (unsigned short)(pos0 + 3);
                wcnt = b3 + __popcll(m3);
                // the LDS operations of one wave complete in order; the fence keeps the compiler from moving the pops above the pushes
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                while (wcnt >= 64 && abl != 2) { wcnt -= 64; score_at(wsurv[wcnt + lane]); }
                if (abl == 2) wcnt = 0;                            // ablation: no scoring
            }
        }
        if (lane < wcnt && abl != 2) score_at(wsurv[lane]);
    }
    __syncthreads();
    FC_STAMP(2);
    // ---- NMS on the list of scored pixels.  With minThFAST <= iniThFAST every stored score is >= minThFAST, so
    // "keep at threshold th" == score >= th and strictly greater than every neighbour inside the cell's rectangle.
    if (abl == 3) return;                                    // no NMS
    const int nnz = nsurv[1];
    const int rh = bh - 6;
    bool listed = g.min_th <= g.ini_th && nnz <= FC_NZ && bncells <= ORB_BAND_CELLS;
    if (listed) {
        // the local maxima go to one bucket of `kept` per cell, so that the row-major rank of a keypoint inside its
        // cell only scans the maxima of that cell (a fifth of the band's on the EuRoC-shaped frames)
        const int cap_c = ((FC_KEPT / max(bncells, 1)) - 1) | 1;      // odd: the buckets start in different LDS banks
        const unsigned p_inv = 0xFFFFFFFFu / (unsigned)P + 1u;           // pos / P == mulhi(pos, p_inv) for pos < 2^16 and every pitch 16..256 (checked exhaustively)
        for (int e = tid; e < nnz; e += FC_TPB) {
            const int pos = nz[e];
            const int row = (int)__umulhi((unsigned)pos, p_inv), col = pos - row * P;
            const int ci = colcell[col];
            if (ci == 255) continue;
            const int xx = col - l_clo[ci], yy = row - 3, rw = l_cwd[ci];
            const uint8_t* p = S + pos;
            const int sc = p[0];
            const bool l = xx > 0, r = xx + 1 < rw, u = yy > 0, d = yy + 1 < rh;
#define NB(c, o) ((c) ? (int)p[o] : 0)
            const bool keep = sc > NB(l, -1) && sc > NB(r, 1) &&
                              sc > NB(u && l, -P - 1) && sc > NB(u, -P) && sc > NB(u && r, -P + 1) &&
                              sc > NB(d && l, P - 1) && sc > NB(d, P) && sc > NB(d && r, P + 1);
#undef NB
            if (keep) {
                const int q = atomicAdd(cell_k + ci, 1);
                if (q < cap_c) kept[ci * cap_c + q] = ((unsigned)yy << 14) | ((unsigned)xx << 8) | (unsigned)sc;
                if (sc >= g.ini_th) atomicAdd(cell_hi + ci, 1);
            }
        }
        __syncthreads();
        FC_STAMP(3);
        // (cell_k of the cells the band does not have is 0)
        const int4 ck = *reinterpret_cast<const int4*>(cell_k);
        const int e1 = ck.x, e2 = e1 + ck.y, e3 = e2 + ck.z, nk = e3 + ck.w;
        listed = max(max(ck.x, ck.y), max(ck.z, ck.w)) <= cap_c;
        if (listed) {
            for (int e = tid; e < nk; e += FC_TPB) {
                const int ci = (e >= e1 ? 1 : 0) + (e >= e2 ? 1 : 0) + (e >= e3 ? 1 : 0);
                const int k0 = e - (ci == 0 ? 0 : ci == 1 ? e1 : ci == 2 ? e2 : e3);
                const unsigned* bucket = kept + ci * cap_c;
                const int nb = ci == 0 ? ck.x : ci == 1 ? ck.y : ci == 2 ? ck.z : ck.w;
                const unsigned me = bucket[k0];
                const unsigned th = (unsigned)(cell_hi[ci] > 0 ? g.ini_th : g.min_th);      // :978-984: ini first, else min
                if ((me & 255u) < th) continue;
                int rank = 0;                                                             // row-major order inside the cell
                for (int k = 0; k < nb; k++) {
                    const unsigned o = bucket[k];
                    rank += ((o & 255u) >= th && (o >> 8) < (me >> 8)) ? 1 : 0;
                }
                if (rank < l_scap[ci])
                    slots[(long long)f * g.slots_per_frame + l_sfirst[ci] + rank] =
                        ((me & 255u) << 24) | ((unsigned)(by0 + 3 - ORB_BORDER + (int)((me >> 14) & 63u)) << 12) |
                        (unsigned)(bxa + l_clo[ci] - ORB_BORDER + (int)((me >> 8) & 63u));
            }
            // a cell's count = its local maxima at the chosen threshold: one thread per cell counts its bucket (an atomic per keypoint
            // and a barrier before the store, until round 3)
            if (tid >= FC_TPB - FC_CELLS && FC_TPB - 1 - tid < bncells) {
                const int ci = FC_TPB - 1 - tid;
                const unsigned th = (unsigned)(cell_hi[ci] > 0 ? g.ini_th : g.min_th);
                const unsigned* bucket = kept + ci * cap_c;
                int cnt = 0;
                for (int k = 0; k < cell_k[ci]; k++) cnt += (bucket[k] & 255u) >= th ? 1 : 0;
                cell_count[(long long)f * g.ncells + bcfirst + ci] = cnt;
            }
            FC_STAMP(4);
            return;
        }
    }
    // ---- per cell: threshold choice, strict 3x3 NMS inside the cell's detection rectangle, ordered compaction
    for (int ci = wv; ci < B.ncells; ci += FC_TPB / 64) {
        const OrbCell c = cells[B.cell_first + ci];
        const int rw = c.cw - 6, rh = c.ch - 6;
        int* count_out = cell_count + (long long)f * g.ncells + B.cell_first + ci;
        if (rw <= 0 || rh <= 0) { if (lane == 0) *count_out = 0; continue; }
        const uint8_t* S0 = S + 3 * P + (c.x0 + 3 - B.xa);                    // score of the rectangle's first pixel
        unsigned* out = slots + (long long)f * g.slots_per_frame + c.slot_first;
        const int relx = c.x0 + 3 - ORB_BORDER, rely = c.y0 + 3 - ORB_BORDER;
        const int rpi = rw <= 32 ? 2 : 1;
        const int xx = rpi == 2 ? (lane & 31) : lane, yoff = rpi == 2 ? (lane >> 5) : 0;
        int n = 0;
        for (int attempt = 0; attempt < 2 && n == 0; attempt++) {
            const int th = attempt == 0 ? g.ini_th : g.min_th;
            for (int y0 = 0; y0 < rh; y0 += rpi) {
                const int yy = y0 + yoff;
                const bool in = yy < rh && xx < rw;
                const uint8_t* p = S0 + min(yy, rh - 1) * P + min(xx, rw - 1);
                const int s = in ? p[0] : 0;
                if (__ballot(s >= th && s > 0) == 0ull) continue;              // most rows hold no corner
                bool keep = false;
                if (s >= th && s > 0) {
                    // neighbours outside the cell's rectangle, and scores below the threshold, count as 0
                    const bool l = xx > 0, r = xx + 1 < rw, u = yy > 0, d = yy + 1 < rh;
#define NB(c, o) ((c) && (int)p[o] >= th ? (int)p[o] : 0)
                    keep = s > NB(l, -1) && s > NB(r, 1) &&
                           s > NB(u && l, -P - 1) && s > NB(u, -P) && s > NB(u && r, -P + 1) &&
                           s > NB(d && l, P - 1) && s > NB(d, P) && s > NB(d && r, P + 1);
#undef NB
                }
                const unsigned long long m = __ballot(keep);
                if (keep) {
                    const int pos = n + __popcll(m & lanemask_lt());
                    if (pos < c.slot_cap)
                        out[pos] = ((unsigned)s << 24) | ((unsigned)(rely + yy) << 12) | (unsigned)(relx + xx);
                }
                n += __popcll(m);
            }
        }
        if (lane == 0) *count_out = n;
    }
}

// ------------------------------------------------------------------------------------------------
// k_octree: DistributeOctTree (ORBextractor.cpp:707-931), one workgroup of 4 waves per (frame, level): the loops over
// keys are dealt to the threads (see KEY-PARALLEL FORM in the kernel), the list bookkeeping is computed by every wave
// identically (same LDS values written four times) so that all loop control stays uniform across the block.
//
// The reference keeps nodes in a std::list, inserting children with push_front and erasing the
// parent.  Equivalent array form used here: after a pass in which the nodes P_0..P_{m-1} are divided
// in processing order, the new list is   reverse(children in creation order) ++ (old list minus the
// divided nodes).  The keys are not moved (rounds 1-2 kept them grouped by node in a ping-pong scratch array, a stable 4-way
// partition per divided node: DivideNode pushes keys in order, :681-695): every key carries the list position of its node.
// "Full" passes divide every node holding more than one key, in list order (:774-833).  Once
// size + 3*nToExpand > N the reference switches to dividing in descending (size, address) order and
// stops as soon as size >= N (:841-905); equal sizes are ordered by creation (later first), the
// same deterministic replacement for the address order that the oracle documents.
struct OctNodes {
    int* first; int* count;
    short* x0; short* y0; short* x1; short* y1;
    uint8_t* buf;
};
#define OCT_SET_BYTES 20       // per node in one OctNodes set (17 used)
#define OCT_NODE_LDS 72        // total LDS bytes per list slot: 2 sets + cc[4] + ord + cbase + mark + gain

__device__ __forceinline__ OctNodes oct_carve(char* base, int cap)
{
    OctNodes n;
    n.first = reinterpret_cast<int*>(base);              base += 4 * cap;
    n.count = reinterpret_cast<int*>(base);              base += 4 * cap;
    n.x0 = reinterpret_cast<short*>(base);               base += 2 * cap;
    n.y0 = reinterpret_cast<short*>(base);               base += 2 * cap;
    n.x1 = reinterpret_cast<short*>(base);               base += 2 * cap;
    n.y1 = reinterpret_cast<short*>(base);               base += 2 * cap;
    n.buf = reinterpret_cast<uint8_t*>(base);
    return n;
}

__device__ __forceinline__ int key_x(unsigned k) { return (int)(k & 0xFFFu); }
__device__ __forceinline__ int key_y(unsigned k) { return (int)((k >> 12) & 0xFFFu); }

// Orders the LDS and global accesses of the block's 4 waves between phases.
__device__ __forceinline__ void wave_sync_mem()
{
    __threadfence_block();
    __syncthreads();
}

__device__ __forceinline__ int oct_nonempty(const int* cc, int p)
{
    return (cc[4 * p] > 0) + (cc[4 * p + 1] > 0) + (cc[4 * p + 2] > 0) + (cc[4 * p + 3] > 0);
}

// Diagnostic build only (-DOCT_STAMPS, tools/r03_oct_stamps.sh): the level-0 workgroup of frame 0 leaves the shader clock at its phase
// boundaries; the shipped kernel contains none of this.
#ifdef OCT_STAMPS
__device__ unsigned long long oct_stamps[64];
__device__ int oct_stamp_n;
#define OCT_STAMP() do { if (threadIdx.x == 0 && blockIdx.x == 0 && blockIdx.y == 0) { const int i__ = oct_stamp_n; if (i__ < 64) { oct_stamps[i__] = __builtin_amdgcn_s_memtime(); oct_stamp_n = i__ + 1; } } } while (0)
extern "C" int ccm_debug_oct_stamps(unsigned long long* out, int* n)
{
    if (hipMemcpyFromSymbol(n, HIP_SYMBOL(oct_stamp_n), 4, 0, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    const int zero = 0;
    if (hipMemcpyFromSymbol(out, HIP_SYMBOL(oct_stamps), 64 * 8, 0, hipMemcpyDeviceToHost) != hipSuccess) return -1;
    return hipMemcpyToSymbol(HIP_SYMBOL(oct_stamp_n), &zero, 4, 0, hipMemcpyHostToDevice) == hipSuccess ? 0 : -1;
}
#else
#define OCT_STAMP() do { } while (0)
#endif
#ifndef OCT_LDS_KEYS
#define OCT_LDS_KEYS 2048       // candidates of a (frame, level) up to which the key-parallel form below is used (its keys and their node ids live in LDS)
#endif
#ifndef OCT_WAVES
#define OCT_WAVES 4             // measured: 1 -> 0.22 ms, 2 -> 0.41, 4 -> 0.135, 8 -> 0.23 ms per 256 frames
#endif
// (Round 3 also put the two key buffers into LDS for levels with <= 2048 candidates -- every level of the bench frames -- so that the
//  passes below chain LDS instead of global round trips: 0.1208 ms against 0.1198 for this kernel, no gain; the passes are bound by
//  their barriers and the list bookkeeping every wave repeats, not by where the keys live.  Not kept.)
// (90 registers = 5 workgroups per CU for 2048 workgroups; forcing 6 / 8 with -DOCT_WPE (40 / 96 bytes of spills) measured no gain:
//  step 1.064 / 1.071 / 1.083 ms at 5 / 6 / 8 -- the kernel is the chain inside a workgroup, not the second round of workgroups)
#ifndef OCT_WPE
#define OCT_WPE 5                // 96 registers: five workgroups per CU
#endif
#define OCT_OCC __attribute__((amdgpu_waves_per_eu(OCT_WPE, OCT_WPE)))
__global__ __launch_bounds__(64 * OCT_WAVES) OCT_OCC void k_octree(const OrbGeom g, const OrbCell* __restrict__ cells,
                                               const unsigned* __restrict__ slots, const int* __restrict__ cell_count,
                                               unsigned* keysA, unsigned* keysB,
                                               unsigned* __restrict__ out, int* __restrict__ out_count,
                                               int* __restrict__ status, int reg_keys_on)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    OCT_STAMP();                                             // 0 start
    // blockIdx.y = level: the long-running fine levels are dispatched first
    const int level = blockIdx.y, f = blockIdx.x + g.frame0, lane = lane_id(), wv = threadIdx.x >> 6;
    const OrbLevel& L = g.lv[level];
    const int cap = g.list_cap;                                          // multiple of 16
    OctNodes cur = oct_carve(smem, cap);
    OctNodes nxt = oct_carve(smem + (size_t)cap * OCT_SET_BYTES, cap);
    int* cc = reinterpret_cast<int*>(smem + (size_t)cap * 2 * OCT_SET_BYTES);   // [cap][4] child key counts
    int* ord = cc + 4 * cap;          // processing order -> list position
    int* cbase = ord + cap;           // creation index of the first child of the k-th divided node
    int* mark = cbase + cap;          // 0 = survivor, k+1 = divided as the k-th
    int* gain = mark + cap;           // careful mode: list growth per candidate, in rank order
    int* shared_len = gain + cap;     // one uniform word

    unsigned* kb[2] = { keysA + (long long)f * g.keys_per_frame + L.key_first,
                        keysB + (long long)f * g.keys_per_frame + L.key_first };
    // KEY-PARALLEL FORM (round 3).  The reference moves
    // a divided node's keys into its children's vectors, and until round 3 this kernel did the same: a stable 4-way partition of
    // every divided node's key range, a wave per node -- half of the kernel's time, most lanes idle on nodes of ~10 keys
    // (tools/r03_oct_stamps.sh).  But a key's position inside its node never matters: DivideNode keeps the keys in their original
    // order, and the only thing read from a node's key list besides its length is "the best response, first key wins ties"
    // (:912-928).  So the keys stay where the gather put them, every key carries the list position of its node, and a pass is
    // a loop over KEYS: count the quadrants of the nodes that may be divided (LDS atomics: sums, order-free), and after the list
    // bookkeeping (unchanged: it never looks at keys) move every key's node id to its child's or its survivor's new position.
    // The final choice is an atomic max per node of (response, original order reversed).
    unsigned* lkeys = reinterpret_cast<unsigned*>(shared_len + 4);                     // [OCT_LDS_KEYS]
    unsigned* best = lkeys + OCT_LDS_KEYS;                                               // [cap]
    unsigned short* surv_pos = reinterpret_cast<unsigned short*>(best + cap);          // [cap] new position of a node that is not divided
    const int N = L.quota;
    int* ocount = out_count + (long long)f * g.nlevels + level;

    // ---- gather this level's candidates in cell-major order into kb[0]: 64 cells per wave and round
    const int* ccount = cell_count + (long long)f * g.ncells + L.cell_first;
    const unsigned* fslots = slots + (long long)f * g.slots_per_frame;
    int total = 0;
    for (int base = 0; base < L.ncells; base += 64 * OCT_WAVES) {
        int cnt[OCT_WAVES];
#pragma unroll
        for (int u = 0; u < OCT_WAVES; u++) {
            const int ci = base + 64 * u + lane;
            cnt[u] = ci < L.ncells ? ccount[ci] : 0;
        }
        const int myci = base + 64 * wv + lane;
        const int sfirst = myci < L.ncells ? cells[L.cell_first + myci].slot_first : 0;
        int mydst = 0, mycnt = 0;
#pragma unroll
        for (int u = 0; u < OCT_WAVES; u++) {
            const int incl = wave_incl_scan(cnt[u]);
            if (u == wv) { mydst = total + incl - cnt[u]; mycnt = cnt[u]; }
            total += __shfl(incl, 63, 64);
        }
        OCT_STAMP();                                         // (diagnostic: counts and slot offsets known)
        const unsigned* sp = fslots + sfirst;
        // (eight loads in flight per cell -- a cell of the bench frames holds up to ~20 candidates: with four per trip this loop was
        //  five dependent global round trips, twice: 30,000 of the kernel's 115,000 cycles, tools/r03_oct_stamps.sh; sixteen cost the
        //  registers that keep five workgroups on a CU)
        for (int k = 0; k < mycnt; k += 8) {
            unsigned v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) v[u] = sp[min(k + u, mycnt - 1)];
#pragma unroll
            for (int u = 0; u < 8; u++)
                if (k + u < mycnt && mydst + k + u < L.key_cap) {
                    kb[0][mydst + k + u] = v[u];
                    if (mydst + k + u < OCT_LDS_KEYS) lkeys[mydst + k + u] = v[u];
                }
        }
    }
    OCT_STAMP();                                             // (diagnostic: this wave's keys stored)
    if (total > L.key_cap) { if (lane == 0) atomicOr(status, 1); total = L.key_cap; }
    const int n = total;
    if (n == 0 || N <= 0) { if (threadIdx.x == 0) *ocount = 0; return; }
    const bool small = reg_keys_on && n <= OCT_LDS_KEYS;
    wave_sync_mem();
    OCT_STAMP();                                             // 1 candidates gathered

    // ---- the loops over keys.  Up to OCT_LDS_KEYS candidates: the thread's keys (k = tid + 256 j) and their nodes' list positions
    //      stay in REGISTERS through all passes (unrolled: the LDS reads of a thread's eight keys are independent and in flight
    //      together).  More: the keys are read from kb[0] and the node ids live in kb[1] (thread t only ever touches the ids of its
    //      own keys, so no barrier is needed for them).
    const int tid = threadIdx.x;
    const int R = L.roots;
    constexpr int OCT_KPT = OCT_LDS_KEYS / (64 * OCT_WAVES);
    unsigned mykey[OCT_KPT]; int mynode[OCT_KPT];
#pragma unroll
    for (int j = 0; j < OCT_KPT; j++) { mykey[j] = 0u; mynode[j] = 0; }
    if (small) {
#pragma unroll
        for (int j = 0; j < OCT_KPT; j++) { const int k = tid + 64 * OCT_WAVES * j; if (k < n) mykey[j] = lkeys[k]; }
    }
    // fn(key, node, k); mode 0: node read, 1: node read and written, 2: node written
    auto for_keys = [&](auto&& fn, int mode) {
        if (small) {
#pragma unroll
            for (int j = 0; j < OCT_KPT; j++) { const int k = tid + 64 * OCT_WAVES * j; if (k < n) fn(mykey[j], mynode[j], k); }
        } else {
            for (int k = tid; k < n; k += 64 * OCT_WAVES) {
                const unsigned key = kb[0][k];
                int node = mode == 2 ? 0 : (int)kb[1][k];
                fn(key, node, k);
                if (mode != 0) kb[1][k] = (unsigned)node;
            }
        }
    };
    // ---- roots (:711-753): every key's root from its x, the roots' sizes by LDS atomics, empty roots dropped (list order = x order)
    if (tid < ORB_MAX_ROOTS) cc[tid] = 0;
    __syncthreads();
    for_keys([&](unsigned key, int& node, int) {
        int r = (int)((float)key_x(key) / L.hx); r = min(r, R - 1);
        node = r;
        atomicAdd(&cc[r], 1);
    }, 2);
    __syncthreads();
    if (tid == 0) {
        int p = 0;
        for (int r = 0; r < R; r++) {
            gain[r] = p;                                       // root r's list position (an empty root has no key that would ask)
            if (cc[r] == 0) continue;
            cur.x0[p] = (short)(int)(L.hx * (float)r);
            cur.x1[p] = (short)(int)(L.hx * (float)(r + 1));
            cur.y0[p] = 0; cur.y1[p] = (short)L.bh;
            cur.count[p] = cc[r];
            p++;
        }
        *shared_len = p;
    }
    __syncthreads();
    for_keys([&](unsigned, int& node, int) { node = gain[node]; }, 1);
    wave_sync_mem();
    int len = *shared_len;

    // ---- refinement passes
    OCT_STAMP();                                             // 2 roots made
    bool finish = false, careful = false;
    int guard = 0;
    const unsigned long long lt = lanemask_lt();
    while (!finish) {
        if (++guard > 512) { if (lane == 0) atomicOr(status, 2); break; }
        const int prev = len;
        // (1) the nodes holding more than one key, in list order: the full passes divide exactly these, in this
        //     order.  Every wave builds the whole list and reads back only what it wrote itself.
        int nc = 0;
        for (int base = 0; base < len; base += 64) {
            const int p = base + lane;
            const bool dv = p < len && cur.count[p] > 1;
            const unsigned long long m = __ballot(dv);
            if (dv) ord[nc + __popcll(m & lt)] = p;
            nc += __popcll(m);
        }
        // (2) how many keys of every such node fall into each of its quadrants (n1..n4 of DivideNode, :684-694)
        for (int i = tid; i < 4 * len; i += 64 * OCT_WAVES) cc[i] = 0;
        __syncthreads();
        for_keys([&](unsigned key, int& node, int) {
            const int p = node;
            if (cur.count[p] > 1) {
                const int csx = cur.x0[p] + (cur.x1[p] - cur.x0[p] + 1) / 2, csy = cur.y0[p] + (cur.y1[p] - cur.y0[p] + 1) / 2;   // x0 + ceil(w/2)  (:652)
                atomicAdd(&cc[4 * p + (key_x(key) < csx ? 0 : 1) + (key_y(key) < csy ? 0 : 2)], 1);
            }
        }, 0);
        wave_sync_mem();
        OCT_STAMP();                                         // pass: quadrants counted
        // (3) the divided nodes in processing order: ord[k] = list position of the k-th
        int nd = nc;
        if (careful) {
            // descending (count, creation): new children sit at the list front in reverse creation
            // order, so "created later" == smaller position
            // (the quadratic rank loop is most of this kernel's instructions: each wave ranks a quarter of the list)
            const int m_c = nc;
            // (eight LDS reads in flight: one per step was 145 cycles a step, 17,500 of the kernel's 115,000 cycles; the counts in
            //  registers read back with v_readlane measured slower still)
            for (int base = 64 * wv; base < len; base += 64 * OCT_WAVES) {
                const int p = base + lane;
                const int myc = p < len ? cur.count[p] : 0;
                if (myc > 1) {
                    int rank = 0;
                    for (int q0 = 0; q0 < len; q0 += 8) {
                        int cq[8];
#pragma unroll
                        for (int u = 0; u < 8; u++) cq[u] = cur.count[min(q0 + u, len - 1)];
#pragma unroll
                        for (int u = 0; u < 8; u++) {
                            const int q = q0 + u;
                            rank += (q < len && cq[u] > 1 && (cq[u] > myc || (cq[u] == myc && q < p))) ? 1 : 0;
                        }
                    }
                    ord[rank] = p;
                    gain[rank] = oct_nonempty(cc, p) - 1;
                }
            }
            wave_sync_mem();
            // stop after the first node that brings the list to N (:898); otherwise divide all
            int run = len, K = m_c - 1;
            bool found = false;
            for (int base = 0; base < m_c && !found; base += 64) {
                const int k = base + lane;
                const int gn = k < m_c ? gain[k] : 0;
                const int incl = wave_incl_scan(gn);
                const unsigned long long hit = __ballot(k < m_c && run + incl >= N);
                if (hit != 0ull) { K = base + __ffsll((long long)hit) - 1; found = true; }
                run += __shfl(incl, 63, 64);
            }
            nd = m_c > 0 ? K + 1 : 0;
        }
        OCT_STAMP();                                         // pass: order chosen
        // (4) creation index of each divided node's first child; T = children created in this pass
        int T = 0;
        for (int base = 0; base < nd; base += 64) {
            const int k = base + lane;
            const int ne = k < nd ? oct_nonempty(cc, ord[k]) : 0;
            const int incl = wave_incl_scan(ne);
            if (k < nd) cbase[k] = T + incl - ne;
            T += __shfl(incl, 63, 64);
        }
        // every wave writes all of mark, zeros first: after the barrier each word holds its final value
        for (int base = 0; base < len; base += 64) { const int p = base + lane; if (p < len) mark[p] = 0; }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        for (int base = 0; base < nd; base += 64) { const int k = base + lane; if (k < nd) mark[ord[k]] = k + 1; }
        wave_sync_mem();
        // (5) survivors keep their order behind the new children
        int nsurv = 0;
        for (int base = 0; base < len; base += 64) {
            const int p = base + lane;
            const bool sv = p < len && mark[p] == 0;
            const unsigned long long m = __ballot(sv);
            if (sv) {
                const int dest = T + nsurv + __popcll(m & lt);
                if (dest < cap) {
                    nxt.x0[dest] = cur.x0[p]; nxt.y0[dest] = cur.y0[p]; nxt.x1[dest] = cur.x1[p]; nxt.y1[dest] = cur.y1[p];
                    nxt.count[dest] = cur.count[p];
                    surv_pos[p] = (unsigned short)dest;
                }
            }
            nsurv += __popcll(m);
        }
        const int newlen = T + nsurv;
        if (newlen > cap) { if (lane == 0) atomicOr(status, 4); break; }
        // (6) children, pushed to the front one by one: creation index c -> position T-1-c
        int newExpand = 0;
        for (int base = 0; base < nd; base += 64) {
            const int k = base + lane;
            int gt1 = 0;
            if (k < nd) {
                const int p = ord[k];
                const int px0 = cur.x0[p], py0 = cur.y0[p], px1 = cur.x1[p], py1 = cur.y1[p];
                const int sx = px0 + (px1 - px0 + 1) / 2, sy = py0 + (py1 - py0 + 1) / 2;
                int ci = cbase[k];
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int c = cc[4 * p + q];
                    if (c > 0) {
                        const int dest = T - 1 - ci;
                        nxt.x0[dest] = (short)((q & 1) ? sx : px0);
                        nxt.x1[dest] = (short)((q & 1) ? px1 : sx);
                        nxt.y0[dest] = (short)((q & 2) ? sy : py0);
                        nxt.y1[dest] = (short)((q & 2) ? py1 : sy);
                        nxt.count[dest] = c;
                        ci++;
                        gt1 += (c > 1);
                    }
                }
            }
            newExpand += wave_sum(gt1);
        }
        wave_sync_mem();
        // (7) every key follows its node: to the survivor's new position, or to the child of its quadrant (children are created in
        //     quadrant order, empty ones skipped: creation index c -> position T - 1 - c)
        for_keys([&](unsigned key, int& node, int) {
            const int p = node;
            const int mk = mark[p];
            if (mk == 0) node = surv_pos[p];
            else {
                const int csx = cur.x0[p] + (cur.x1[p] - cur.x0[p] + 1) / 2, csy = cur.y0[p] + (cur.y1[p] - cur.y0[p] + 1) / 2;
                const int q = (key_x(key) < csx ? 0 : 1) + (key_y(key) < csy ? 0 : 2);
                int ci = cbase[mk - 1];
                if (q > 0) ci += cc[4 * p] > 0;
                if (q > 1) ci += cc[4 * p + 1] > 0;
                if (q > 2) ci += cc[4 * p + 2] > 0;
                node = T - 1 - ci;
            }
        }, 1);
        __syncthreads();                                     // (these reads of the old list against the next pass's writes into it)
        { OctNodes t = cur; cur = nxt; nxt = t; }
        len = newlen;
        OCT_STAMP();                                         // pass: new list built
        // (8) loop control (:837-905)
        if (len >= N || len == prev) finish = true;
        else if (!careful && len + 3 * newExpand > N) careful = true;
    }

    // ---- keep the best response of every node, first key wins ties (:912-928); order = list order:
    //      max of response << 24 | (0xFFFFFF - original index)   (at most 2^24 candidates per level: 4.2 M at the largest image)
    unsigned* o = out + (long long)f * g.out_per_frame + L.out_first;
    for (int p = tid; p < len; p += 64 * OCT_WAVES) best[p] = 0u;
    __syncthreads();
    for_keys([&](unsigned key, int& node, int k) { atomicMax(&best[node], (key & 0xFF000000u) | (0xFFFFFFu - (unsigned)k)); }, 0);
    __syncthreads();
    for (int p = tid; p < len && p < L.out_cap; p += 64 * OCT_WAVES) {
        const unsigned kidx = 0xFFFFFFu - (best[p] & 0xFFFFFFu);
        o[p] = small ? lkeys[kidx] : kb[0][kidx];
    }
    if (tid == 0) *ocount = min(len, L.out_cap);
    OCT_STAMP();                                             // end
}

// ------------------------------------------------------------------------------------------------
// k_orient_desc: a wave per selected keypoint (small batches) or per OD_ITEMS keypoint slots (see the kernel).
//   IC_Angle (ORBextractor.cpp:68-95) on the un-blurred level, cv::fastAtan2;
//   GaussianBlur 7x7 sigma 2 BORDER_REFLECT_101 (:1259; integer taps {18,34,49,55,49,34,18},
//   SURVEY.md 12.6) evaluated only on the 37x37 neighbourhood the pattern can reach -- the sum
//   sum_ij q_i q_j src is exact in integers, so it equals the reference's full-image separable
//   blur bit for bit and the blurred pyramid is never written to HBM;
//   computeOrbDescriptor (:100-316): bit k = I(p_2k) < I(p_2k+1) with the pattern rotated by the angle.
__device__ __forceinline__ int reflect101(int i, int n)
{
    if (i < 0) i = -i;
    if (i >= n) i = 2 * (n - 1) - i;
    return min(max(i, 0), n - 1);
}

__device__ __forceinline__ float fast_atan2_deg(float y, float x)
{
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    const float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + 2.2204460492503131e-16f);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + 2.2204460492503131e-16f);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

// Per-wave LDS: hT[37][46] u16 (horizontally blurred, transposed: column-major so the vertical pass reads
// a column as 22 consecutive dwords; 92-byte pitch = 23 dwords is odd, hence bank-conflict free) and
// bl[37][40] u8 (blurred neighbourhood).  bl OVERLAYS hT: the vertical pass has every column in registers
// before its first store (one wave in lockstep, LDS operations of a wave complete in order), so the
// workgroup needs 15 KB instead of 21 KB and 8 of them (32 waves) share a CU.
#define OD_HT_PITCH 92
#define OD_B_PITCH 40
#define OD_WAVE_LDS (ORB_BLUR_D * OD_HT_PITCH)      // 3404 >= 37 * 40
#define OD_WAVE_LDS_PAD ((OD_WAVE_LDS + 15) & ~15)

__constant__ signed char c_pattern[1024];

typedef unsigned short od_us2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned od_dot2(unsigned a, unsigned w, unsigned acc)
{
    return __builtin_amdgcn_udot2(__builtin_bit_cast(od_us2, a), __builtin_bit_cast(od_us2, w), acc, false);
}

#ifndef OD_WAVES
#define OD_WAVES 4               // waves per workgroup
#endif
#ifndef OD_ITEMS
#define OD_ITEMS 8               // keypoint slots per wave (2 / 4 / 8 measured: step 1.070 / 1.061 / 1.049 ms, from 1.125 with one)
#endif
// A wave works through OD_ITEMS slots.  Until round 3 it took one and ended: every keypoint then paid the workgroup launch, the
// weight tables, and the chain slot -> key -> level record -> 11 row loads before its first useful instruction, ~3,500 cycles of a
// lifetime of 23,000 in which the wave held one of the CU's 32 wave slots (profiles/r03_occupancy_sweep.txt: the kernel issues for
// 0.56 of its time at full occupancy).  Now the slots of a wave are decoded together (one lane per slot), and the patch rows of the
// NEXT keypoint are requested as soon as the horizontal blur has consumed the current ones -- into the same registers -- so they
// arrive under the vertical blur and the descriptor.
struct OdItem { const uint8_t* img; int pitch, w, h, cx, cy; };
__device__ __forceinline__ void od_load_patch(const OdItem& it, int lane, unsigned (&prow)[11])
{
    const int gy = reflect101(it.cy + min(lane, ORB_PATCH_D - 1) - ORB_PATCH_R, it.h);
    const uint8_t* rp = it.img + (long long)gy * it.pitch;
    if (it.cx - ORB_PATCH_R >= 0 && it.cx + ORB_PATCH_R + 1 < it.w) {     // wave-uniform: patch inside the row
        const uint8_t* p0 = rp + it.cx - ORB_PATCH_R;
#pragma unroll
        for (int i = 0; i < 11; i++) __builtin_memcpy(&prow[i], p0 + 4 * i, 4);   // unaligned dword loads
    } else {                                                               // BORDER_REFLECT_101 columns
#pragma unroll
        for (int i = 0; i < 11; i++) {
            unsigned v = 0;
#pragma unroll
            for (int j = 0; j < 4; j++) v |= (unsigned)rp[reflect101(it.cx + 4 * i + j - ORB_PATCH_R, it.w)] << (8 * j);
            prow[i] = v;
        }
    }
}
template <int ITEMS>
__global__ __launch_bounds__(64 * OD_WAVES) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_orient_desc(const OrbGeom g, const unsigned* __restrict__ sel,
                                                     const int* __restrict__ sel_count,
                                                     ccm_keypoint* __restrict__ kps, uint8_t* __restrict__ desc,
                                                     int* __restrict__ counts, int max_per_image, int* __restrict__ status, int xcd_on)
{
    extern __shared__ __attribute__((aligned(16))) uint8_t od_pad[];      // (occupancy experiment only: CCM_OD_LDS_PAD)
    __shared__ __attribute__((aligned(16))) uint8_t lds[OD_WAVES][OD_WAVE_LDS_PAD];
    // IC_Angle weights per |v| and patch dword k (columns 4k..4k+3, u = column - 21):
    //   wone = 1 inside the disc row, wu = u + 16 inside (so that sum u*I = dot(wu) - 16*dot(wone) stays unsigned)
    __shared__ unsigned wone[16][11], wu[16][11], pat[256];
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = lane_id();
    int wx, wy;
    xcd_work_item(xcd_on, wx, wy);
    const int f = wy + g.frame0;
    // ---- this wave's slots, one per lane: slot -> (level, k); output row = keypoints of lower levels + k (level-major order, :1249-1276)
    const int* sc = sel_count + (long long)f * g.nlevels;
    const int slot = (wx * ITEMS + min(lane, ITEMS - 1)) * OD_WAVES + wv;
    const unsigned key_l = slot < g.out_per_frame ? sel[(long long)f * g.out_per_frame + slot] : 0u;     // (requested before the tables are built)
    // the 256 test pairs (x0, y0, x1, y1 as signed bytes): read from LDS per keypoint (as global loads they put a wait for ALL of the
    // wave's outstanding memory operations, the previous descriptor stores included, in front of each of the four rounds)
    for (int i = threadIdx.x; i < 256; i += 64 * OD_WAVES) pat[i] = *reinterpret_cast<const unsigned*>(c_pattern + 4 * i);
    for (int i = threadIdx.x; i < 16 * 11; i += 64 * OD_WAVES) {
        const int av = i / 11, k = i - av * 11;
        const int lim = g.umax[av];
        unsigned a = 0, b = 0;
        for (int j = 0; j < 4; j++) {
            const int u = 4 * k + j - ORB_PATCH_R;
            if (u >= -lim && u <= lim) { a |= 1u << (8 * j); b |= (unsigned)(u + 16) << (8 * j); }
        }
        wone[av][k] = a; wu[av][k] = b;
    }
    int level_l = 0, first_l = 0;
    for (int l = 1; l < g.nlevels; l++) if (slot >= g.lv[l].out_first) { level_l = l; first_l = g.lv[l].out_first; }
    int row_l = slot - first_l, tot = 0, cnt_l = 0;
    for (int l = 0; l < g.nlevels; l++) { const int c = sc[l]; if (l < level_l) row_l += c; if (l == level_l) cnt_l = c; tot += c; }
    if (wx == 0 && wv == 0 && lane == 0) {
        counts[f] = min(tot, max_per_image);
        if (tot > max_per_image) atomicOr(status, 8);
    }
    const bool valid_l = lane < ITEMS && slot < g.out_per_frame && slot - first_l < cnt_l && row_l < max_per_image;
    unsigned long long todo = __ballot(valid_l);
    __syncthreads();                                                       // (the weight tables)
    if (todo == 0) return;

    uint8_t* wl = lds[wv];
    uint8_t* bl = wl;                                  // overlays hT, see OD_WAVE_LDS
    const float factorPI = (float)(3.14159265358979323846 / 180.f);
    auto item_of = [&](int j, OdItem& it, int& level, int& score, int& row) {
        const unsigned key = (unsigned)__builtin_amdgcn_readlane((int)key_l, j);
        level = __builtin_amdgcn_readlane(level_l, j); row = __builtin_amdgcn_readlane(row_l, j);
        const OrbLevel& L = g.lv[level];
        it.img = L.img + (long long)f * L.plane; it.pitch = L.pitch; it.w = L.w; it.h = L.h;
        it.cx = (int)(key & 0xFFFu) + ORB_BORDER; it.cy = (int)((key >> 12) & 0xFFFu) + ORB_BORDER;   // :1012-1013
        score = (int)(key >> 24);
    };
    // ---- lane r < 43 holds patch row r (43 pixels + 1 spare byte) in 11 registers
    unsigned prow[11];
    OdItem it; int level, score, row;
    {
        const int j = __builtin_ctzll(todo); todo &= todo - 1;
        item_of(j, it, level, score, row);
        od_load_patch(it, lane, prow);
    }
    for (;;) {
        // ---- IC_Angle: integer moments over the radius-15 disc, one row per lane
        int m10 = 0, m01 = 0;
        {
            const int v = lane - ORB_PATCH_R, av = v < 0 ? -v : v;
            if (av <= ORB_HALF_PATCH) {
                unsigned s1 = 0, su = 0;
#pragma unroll
                for (int i = 1; i < 10; i++) {          // columns 4..39 cover u = -15..15 (columns 6..36)
                    s1 = __builtin_amdgcn_udot4(prow[i], wone[av][i], s1, false);
                    su = __builtin_amdgcn_udot4(prow[i], wu[av][i], su, false);
                }
                m10 = (int)su - 16 * (int)s1;
                m01 = v * (int)s1;
            }
        }
        m10 = wave_sum(m10); m01 = wave_sum(m01);
        const float angle = fast_atan2_deg((float)m01, (float)m10);

        // ---- horizontal 7 taps {18,34,49,55,49,34,18}: two v_dot4_u32_u8 per output, written transposed
        if (lane < ORB_PATCH_D) {
            const unsigned Q0 = 18u | (34u << 8) | (49u << 16) | (55u << 24), Q1 = 49u | (34u << 8) | (18u << 16);
#pragma unroll
            for (int x = 0; x < ORB_BLUR_D; x++) {
                const int kk = x >> 2, sh = x & 3;
                const unsigned w0 = sh ? __builtin_amdgcn_alignbyte(prow[kk + 1], prow[kk], sh) : prow[kk];
                const unsigned w1 = sh ? __builtin_amdgcn_alignbyte(prow[kk + 2 > 10 ? 10 : kk + 2], prow[kk + 1], sh) : prow[kk + 1];
                const unsigned h = __builtin_amdgcn_udot4(w1, Q1, __builtin_amdgcn_udot4(w0, Q0, 0u, false), false);   // <= 65535
                *reinterpret_cast<unsigned short*>(wl + x * OD_HT_PITCH + 2 * lane) = (unsigned short)h;
            }
        }
        // ---- the next keypoint's rows, into the registers the horizontal pass has just finished with
        const bool more = todo != 0;
        const int cx = it.cx, cy = it.cy, level_c = level, score_c = score, row_c = row;
        if (more) {
            const int j = __builtin_ctzll(todo); todo &= todo - 1;
            item_of(j, it, level, score, row);
            od_load_patch(it, lane, prow);
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();
        // ---- vertical 7 taps + the single rounding.  An item is a third of a blurred column (rows 12t .. 12t+12; the
        //      13th row of the first two thirds is also the first of the next: same value, written twice): 111 items in
        //      two rounds of 64 lanes = 26 outputs per lane instead of 37.  Rows are packed two per dword, so four
        //      v_dot2_u32_u16 make one output; a third starts on an even row, so the tap layout is the same for all.
        {
            unsigned d[2][10];
            int wofs[2];
#pragma unroll
            for (int rd = 0; rd < 2; rd++) {
                const int item = min(rd * 64 + lane, 3 * ORB_BLUR_D - 1);
                const int c = (item * 171) >> 9, t = item - 3 * c;                 // item / 3 for item < 128
                const unsigned* col = reinterpret_cast<const unsigned*>(wl + c * OD_HT_PITCH) + 6 * t;
#pragma unroll
                for (int i = 0; i < 10; i++) d[rd][i] = col[i];
                wofs[rd] = 12 * t * OD_B_PITCH + c;
            }
            __builtin_amdgcn_s_waitcnt(0xc07f);                // every column is in registers before bl overwrites hT
            __builtin_amdgcn_wave_barrier();
            asm volatile("" ::: "memory");
#pragma unroll
            for (int rd = 0; rd < 2; rd++) {
                if (rd * 64 + lane < 3 * ORB_BLUR_D) {
#pragma unroll
                    for (int y = 0; y < 13; y++) {
                        const int kk = y >> 1;
                        unsigned v;
                        if ((y & 1) == 0)
                            v = od_dot2(d[rd][kk + 3], 18u, od_dot2(d[rd][kk + 2], 49u | (34u << 16), od_dot2(d[rd][kk + 1], 49u | (55u << 16), od_dot2(d[rd][kk], 18u | (34u << 16), 32768u))));
                        else
                            v = od_dot2(d[rd][kk + 3], 34u | (18u << 16), od_dot2(d[rd][kk + 2], 55u | (49u << 16), od_dot2(d[rd][kk + 1], 34u | (49u << 16), od_dot2(d[rd][kk], 18u << 16, 32768u))));
                        bl[wofs[rd] + y * OD_B_PITCH] = (uint8_t)min(v >> 16, 255u);     // the rounding constant 32768 is the accumulator's start value
                    }
                }
            }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);
        __builtin_amdgcn_wave_barrier();

        // ---- steered BRIEF: lane L evaluates pairs L, 64+L, 128+L, 192+L; a ballot is 8 descriptor bytes
        float a, b;
        ccm_sincosf(angle * factorPI, &b, &a);
        uint8_t* drow = desc + ((long long)f * max_per_image + row_c) * 32;
        const uint8_t* ctr = bl + 18 * OD_B_PITCH + 18;
        unsigned long long mybits = 0;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const unsigned pr = pat[r * 64 + lane];
            const float x0 = (float)(signed char)(pr & 255u), y0 = (float)(signed char)((pr >> 8) & 255u);
            const float x1 = (float)(signed char)((pr >> 16) & 255u), y1 = (float)(signed char)(pr >> 24);
            const int t0 = ctr[__float2int_rn(x0 * b + y0 * a) * OD_B_PITCH + __float2int_rn(x0 * a - y0 * b)];
            const int t1 = ctr[__float2int_rn(x1 * b + y1 * a) * OD_B_PITCH + __float2int_rn(x1 * a - y1 * b)];
            const unsigned long long bits = __ballot(t0 < t1);
            if (lane == r) mybits = bits;
        }
        if (lane < 4) *reinterpret_cast<unsigned long long*>(drow + 8 * lane) = mybits;      // one 32-byte store
        if (lane == 0) {
            const OrbLevel& L = g.lv[level_c];
            ccm_keypoint kp;
            kp.x = (float)cx; kp.y = (float)cy;
            if (level_c != 0) { kp.x *= L.scale; kp.y *= L.scale; }              // :1268-1274
            kp.size = L.kp_size; kp.angle = angle; kp.response = (float)score_c;
            kp.octave = level_c; kp.class_id = -1;
            kps[(long long)f * max_per_image + row_c] = kp;
        }
        if (!more) break;
        // (the BRIEF reads of bl are in registers -- the ballots consumed them -- before the next horizontal pass overwrites it)
    }
}

// ------------------------------------------------------------------------------------------------
// host-side launchers (called from orb_host.cpp)
extern "C" hipError_t orb_upload_pattern()
{
    return hipMemcpyToSymbol(HIP_SYMBOL(c_pattern), ccm_orb_pattern, 1024);
}

size_t orb_octree_lds_bytes(int list_cap) { return (size_t)list_cap * OCT_NODE_LDS + 64 + 4 * (size_t)OCT_LDS_KEYS + 8 * (size_t)list_cap; }

void orb_launch_resize(hipStream_t s, const OrbGeom& g_dev, int level, int dw, int dh, int nframes)
{
    dim3 grid((dw + 255) / 256, (dh + 4 * RS_ROWS - 1) / (4 * RS_ROWS), nframes);
    hipLaunchKernelGGL(k_pyr_resize, grid, dim3(256), 0, s, g_dev, level);
}
void orb_launch_score(hipStream_t s, const OrbGeom& g_dev, int ntiles, int nframes)
{
    hipLaunchKernelGGL(k_fast_score, dim3(ntiles, nframes), dim3(256), 0, s, g_dev);
}
void orb_launch_nms(hipStream_t s, const OrbGeom& g_dev, const OrbCell* cells, int ncells, int nframes,
                    unsigned* slots, int* cell_count)
{
    hipLaunchKernelGGL(k_cell_nms, dim3((ncells + 3) / 4, nframes), dim3(256), 0, s, g_dev, cells, slots, cell_count);
}
void orb_launch_fast_cells(hipStream_t s, const OrbGeom& g_dev, const OrbCell* cells, const OrbBand* bands, int nbands, int nframes,
                           size_t lds_bytes, int surv_cap, unsigned* slots, int* cell_count)
{
    static const bool packed = !(getenv("CCM_FC_PACKED") && atoi(getenv("CCM_FC_PACKED")) == 0);
    static const int abl = getenv("CCM_FC_ABL") ? atoi(getenv("CCM_FC_ABL")) : 0;     // timing ablations only (results are wrong)
    // default 0 for this kernel: co-locating neighbouring bands on one XCD cuts its fabric reads 2.2x (285 -> 128 MiB per launch, = the
    // algorithmic bytes) but the kernel, which is not bandwidth-bound, runs 3-5 % slower with every co-locating order tried
    // (profiles/r02_xcd_order.txt); k_orient_desc gains from it and uses it
    static const int xcd_on = getenv("CCM_ORB_XCD_FC") ? atoi(getenv("CCM_ORB_XCD_FC")) : (getenv("CCM_ORB_XCD") ? atoi(getenv("CCM_ORB_XCD")) : 0);
    // occupancy experiment (tools/r03_occupancy_sweep.sh): a larger LDS request leaves fewer workgroups per CU; results are unchanged
    static const size_t lds_min = getenv("CCM_FC_LDS_MIN") ? (size_t)atol(getenv("CCM_FC_LDS_MIN")) : 0;
    if (lds_bytes < lds_min) lds_bytes = lds_min;
    (void)surv_cap;                                          // (rounds 1-2: size of the pooled survivor list)
    if (packed) hipLaunchKernelGGL(k_fast_cells<true>, dim3(nbands, nframes), dim3(FC_TPB), lds_bytes, s, g_dev, cells, bands, slots, cell_count, 0, abl, xcd_on);
    else hipLaunchKernelGGL(k_fast_cells<false>, dim3(nbands, nframes), dim3(FC_TPB), lds_bytes, s, g_dev, cells, bands, slots, cell_count, 0, abl, xcd_on);
}
size_t orb_fast_cells_lds(int pitch, int bh, int surv_cap) { return fc_lds_bytes(pitch, bh, surv_cap); }
void orb_launch_octree(hipStream_t s, const OrbGeom& g_dev, const OrbCell* cells, int nlevels, int nframes, int list_cap,
                       const unsigned* slots, const int* cell_count, unsigned* keysA, unsigned* keysB,
                       unsigned* out, int* out_count, int* status)
{
    static const int key_parallel = !(getenv("CCM_OCT_REG_KEYS") && atoi(getenv("CCM_OCT_REG_KEYS")) == 0);      // 0: keys and node ids in global memory on every level (test switch: the form levels with > OCT_LDS_KEYS candidates take)
    hipLaunchKernelGGL(k_octree, dim3(nframes, nlevels), dim3(64 * OCT_WAVES), orb_octree_lds_bytes(list_cap), s,
                       g_dev, cells, slots, cell_count, keysA, keysB, out, out_count, status, key_parallel);
}
void orb_launch_orient_desc(hipStream_t s, const OrbGeom& g_dev, int out_per_frame, int nframes, const unsigned* sel,
                            const int* sel_count, ccm_keypoint* kps, uint8_t* desc, int* counts, int max_per_image,
                            int* status)
{
    static const int xcd_on = getenv("CCM_ORB_XCD_OD") ? atoi(getenv("CCM_ORB_XCD_OD")) : (getenv("CCM_ORB_XCD") ? atoi(getenv("CCM_ORB_XCD")) : 1);
    static const size_t lds_pad = getenv("CCM_OD_LDS_PAD") ? (size_t)atol(getenv("CCM_OD_LDS_PAD")) : 0;     // occupancy experiment only
    // a wave per slot while that still fills the GPU only a few times over (a single frame is 1,032 waves: eight slots per wave would be
    // 33 workgroups working through their slots one after the other), OD_ITEMS slots per wave for batches
    if ((long long)out_per_frame * nframes < 4 * 8192)
        hipLaunchKernelGGL(k_orient_desc<1>, dim3((out_per_frame + OD_WAVES - 1) / OD_WAVES, nframes), dim3(64 * OD_WAVES), lds_pad, s,
                           g_dev, sel, sel_count, kps, desc, counts, max_per_image, status, xcd_on);
    else
        hipLaunchKernelGGL(k_orient_desc<OD_ITEMS>, dim3((out_per_frame + OD_WAVES * OD_ITEMS - 1) / (OD_WAVES * OD_ITEMS), nframes), dim3(64 * OD_WAVES), lds_pad, s,
                           g_dev, sel, sel_count, kps, desc, counts, max_per_image, status, xcd_on);
}

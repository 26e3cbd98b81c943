/* orb_oracle.c -- CPU restatement of ORBextractor.  TEST INFRASTRUCTURE (see oracle.h).
 *
 * Follows cslam/src/ORBextractor.cpp of the reference (file:line cited per
 * function) and, for the OpenCV primitives the reference calls but does not
 * contain, the definitions adopted in SURVEY.md section 12 (parity unpinned).
 * Plain scalar C, single thread, like the reference.
 */
#include "oracle.h"
#include "../include/ccm_orb_pattern.h"
#include "../include/ccm_sincos.h"
#include <math.h>
#include <float.h>
#include <stdlib.h>
#include <string.h>

#define PATCH_SIZE 31
#define HALF_PATCH 15
#define EDGE_TH 19

/* cvRound: nearest, ties to even. */
int orc_round_half_even(double v) { return (int)lrint(v); }
static int rnd_f(float v) { return (int)lrintf(v); }

/* ------------------------------------------------------------------ A1: ctor tables
 * ORBextractor.cpp:579-639 */
int orc_orb_tables(const orc_orb_params* p, float* scale, float* inv_scale, float* sigma2,
                   float* inv_sigma2, int32_t* nfeat, int32_t* umax)
{
    if (!p || p->nlevels < 1 || p->nlevels > 16) return -1;
    const int n = p->nlevels;
    float sc[16], s2[16];
    sc[0] = 1.0f; s2[0] = 1.0f;
    for (int i = 1; i < n; i++) {
        sc[i] = sc[i - 1] * p->scale_factor;          /* :590 */
        s2[i] = sc[i] * sc[i];                        /* :591 */
    }
    for (int i = 0; i < n; i++) {
        if (scale) scale[i] = sc[i];
        if (sigma2) sigma2[i] = s2[i];
        if (inv_scale) inv_scale[i] = 1.0f / sc[i];   /* :598 */
        if (inv_sigma2) inv_sigma2[i] = 1.0f / s2[i]; /* :599 */
    }
    if (nfeat) {
        float factor = 1.0f / p->scale_factor;        /* :605 */
        float nd = p->nfeatures * (1 - factor) / (1 - (float)pow((double)factor, (double)n)); /* :606 */
        int sum = 0;
        for (int l = 0; l < n - 1; l++) {
            nfeat[l] = rnd_f(nd);                     /* :611 */
            sum += nfeat[l];
            nd *= factor;
        }
        nfeat[n - 1] = p->nfeatures - sum > 0 ? p->nfeatures - sum : 0; /* :615 */
    }
    if (umax) {
        /* :623-638 circular patch row half-widths */
        int v, v0;
        int vmax = (int)floorf(HALF_PATCH * sqrtf(2.f) / 2 + 1);
        int vmin = (int)ceilf(HALF_PATCH * sqrtf(2.f) / 2);
        const double hp2 = HALF_PATCH * HALF_PATCH;
        for (v = 0; v <= HALF_PATCH; v++) umax[v] = 0;
        for (v = 0; v <= vmax; ++v) umax[v] = orc_round_half_even(sqrt(hp2 - v * v));
        for (v = HALF_PATCH, v0 = 0; v >= vmin; --v) {
            while (umax[v0] == umax[v0 + 1]) ++v0;
            umax[v] = v0;
            ++v0;
        }
    }
    return 0;
}

/* ORBextractor.cpp:1284-1285 */
int orc_orb_level_sizes(const orc_orb_params* p, int w, int h, int32_t* lw, int32_t* lh)
{
    float inv[16];
    if (orc_orb_tables(p, 0, inv, 0, 0, 0, 0)) return -1;
    for (int l = 0; l < p->nlevels; l++) {
        lw[l] = rnd_f((float)w * inv[l]);
        lh[l] = rnd_f((float)h * inv[l]);
    }
    return 0;
}

/* ------------------------------------------------------------------ A2: cv::resize
 * INTER_LINEAR, 8-bit single channel: 11-bit fixed-point weights,
 * horizontal pass to int32, vertical pass with the 8u shift sequence. */
static void lin_coeffs(int ssize, int dsize, int* ofs, short* coef)
{
    const double inv_scale = (double)dsize / ssize;
    const double scale = 1. / inv_scale;
    for (int d = 0; d < dsize; d++) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int s = (int)floorf(f);
        f -= s;
        if (s < 0) { s = 0; f = 0.f; }
        if (s >= ssize - 1) { s = ssize - 1; f = 0.f; }
        ofs[d] = s;
        coef[2 * d] = (short)rnd_f((1.f - f) * 2048.f);
        coef[2 * d + 1] = (short)rnd_f(f * 2048.f);
    }
}

void orc_resize_linear_u8(const uint8_t* src, int sw, int sh, int sstride,
                          uint8_t* dst, int dw, int dh, int dstride)
{
    int* xo = (int*)malloc(sizeof(int) * dw);
    int* yo = (int*)malloc(sizeof(int) * dh);
    short* xa = (short*)malloc(sizeof(short) * 2 * dw);
    short* ya = (short*)malloc(sizeof(short) * 2 * dh);
    int* r0 = (int*)malloc(sizeof(int) * dw);
    int* r1 = (int*)malloc(sizeof(int) * dw);
    lin_coeffs(sw, dw, xo, xa);
    lin_coeffs(sh, dh, yo, ya);
    for (int y = 0; y < dh; y++) {
        int sy0 = yo[y];
        int sy1 = sy0 + 1 < sh ? sy0 + 1 : sh - 1;
        const uint8_t* S0 = src + (size_t)sy0 * sstride;
        const uint8_t* S1 = src + (size_t)sy1 * sstride;
        for (int x = 0; x < dw; x++) {
            int sx0 = xo[x];
            int sx1 = sx0 + 1 < sw ? sx0 + 1 : sw - 1;
            r0[x] = S0[sx0] * xa[2 * x] + S0[sx1] * xa[2 * x + 1];
            r1[x] = S1[sx0] * xa[2 * x] + S1[sx1] * xa[2 * x + 1];
        }
        const int b0 = ya[2 * y], b1 = ya[2 * y + 1];
        uint8_t* D = dst + (size_t)y * dstride;
        for (int x = 0; x < dw; x++)
            D[x] = (uint8_t)((((b0 * (r0[x] >> 4)) >> 16) + ((b1 * (r1[x] >> 4)) >> 16) + 2) >> 2);
    }
    free(xo); free(yo); free(xa); free(ya); free(r0); free(r1);
}

/* ------------------------------------------------------------------ A4: cv::FAST
 * TYPE_9_16 with non-max suppression on ONE (sub-)image. */
static const int ring_dx[16] = { 0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1 };
static const int ring_dy[16] = { 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3 };

/* corner test as defined: more than 8 contiguous ring pixels all brighter than
 * v+t or all darker than v-t */
static int fast_is_corner(const int* ring, int v, int t)
{
    int run_b = 0, run_d = 0;
    for (int k = 0; k < 16 + 8; k++) {     /* 25 positions = 16 + 9 - 1 cover every arc */
        int x = ring[k & 15];
        if (x > v + t) { if (++run_b > 8) return 1; } else run_b = 0;
        if (x < v - t) { if (++run_d > 8) return 1; } else run_d = 0;
    }
    return 0;
}

/* cornerScore<16>: the largest threshold for which the pixel is still a corner */
static int fast_corner_score(const int* ring, int v, int threshold)
{
    int d[25];
    for (int k = 0; k < 25; k++) d[k] = v - ring[k & 15];
    int a0 = threshold;
    for (int k = 0; k < 16; k += 2) {
        int a = d[k + 1] < d[k + 2] ? d[k + 1] : d[k + 2];
        if (d[k + 3] < a) a = d[k + 3];
        if (a <= a0) continue;
        for (int j = 4; j <= 8; j++) if (d[k + j] < a) a = d[k + j];
        int m = a < d[k] ? a : d[k];
        if (m > a0) a0 = m;
        m = a < d[k + 9] ? a : d[k + 9];
        if (m > a0) a0 = m;
    }
    int b0 = -a0;
    for (int k = 0; k < 16; k += 2) {
        int b = d[k + 1] > d[k + 2] ? d[k + 1] : d[k + 2];
        if (d[k + 3] > b) b = d[k + 3];
        if (b >= b0) continue;
        for (int j = 4; j <= 8; j++) if (d[k + j] > b) b = d[k + j];
        int m = b > d[k] ? b : d[k];
        if (m < b0) b0 = m;
        m = b > d[k + 9] ? b : d[k + 9];
        if (m < b0) b0 = m;
    }
    return -b0 - 1;
}

int orc_fast9_16(const uint8_t* img, int w, int h, int stride, int threshold,
                 int32_t* xy, int32_t* score, int max)
{
    if (threshold < 0) threshold = 0;
    if (threshold > 255) threshold = 255;
    if (w < 7 || h < 7) return 0;
    uint8_t* sc = (uint8_t*)calloc((size_t)w * h, 1);     /* score, 0 where not a corner */
    uint8_t* is = (uint8_t*)calloc((size_t)w * h, 1);     /* corner flag */
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            const uint8_t* c = img + (size_t)y * stride + x;
            const int v = c[0], t = threshold;
            /* every 9-arc contains ring pixel k or k+8: if both are within +-t no arc can qualify */
            {
                int a = c[3 * stride], b = c[-3 * stride];
                if (a <= v + t && a >= v - t && b <= v + t && b >= v - t) continue;
                a = c[3]; b = c[-3];
                if (a <= v + t && a >= v - t && b <= v + t && b >= v - t) continue;
            }
            int ring[16];
            for (int k = 0; k < 16; k++) ring[k] = c[ring_dy[k] * stride + ring_dx[k]];
            if (fast_is_corner(ring, v, t)) {
                is[(size_t)y * w + x] = 1;
                sc[(size_t)y * w + x] = (uint8_t)fast_corner_score(ring, v, t);
            }
        }
    int n = 0;
    for (int y = 3; y < h - 3; y++)
        for (int x = 3; x < w - 3; x++) {
            if (!is[(size_t)y * w + x]) continue;
            const uint8_t* c = sc + (size_t)y * w + x;
            const int s = c[0];
            if (s > c[-1] && s > c[1] && s > c[-w - 1] && s > c[-w] && s > c[-w + 1] &&
                s > c[w - 1] && s > c[w] && s > c[w + 1]) {
                if (n < max) { xy[2 * n] = x; xy[2 * n + 1] = y; score[n] = s; }
                n++;
            }
        }
    free(sc); free(is);
    return n;
}

/* ------------------------------------------------------------------ A7: GaussianBlur
 * 7x7, sigma 2, BORDER_REFLECT_101, 8-bit fixed-point path:
 * taps round(g*256) = {18,34,49,55,49,34,18}, one rounding after both passes. */
static int reflect101(int i, int n)
{
    if (n == 1) return 0;
    while (i < 0 || i >= n) { if (i < 0) i = -i; else i = 2 * (n - 1) - i; }
    return i;
}
static const int blur_q[7] = { 18, 34, 49, 55, 49, 34, 18 };

void orc_blur7_sigma2(const uint8_t* src, int w, int h, int sstride, uint8_t* dst, int dstride)
{
    int* hp = (int*)malloc(sizeof(int) * (size_t)w * h);
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int i = 0; i < 7; i++) s += blur_q[i] * src[(size_t)y * sstride + reflect101(x + i - 3, w)];
            hp[(size_t)y * w + x] = s;
        }
    for (int y = 0; y < h; y++)
        for (int x = 0; x < w; x++) {
            int s = 0;
            for (int i = 0; i < 7; i++) s += blur_q[i] * hp[(size_t)reflect101(y + i - 3, h) * w + x];
            s = (s + 32768) >> 16;
            dst[(size_t)y * dstride + x] = (uint8_t)(s > 255 ? 255 : s);
        }
    free(hp);
}

/* ------------------------------------------------------------------ cv::fastAtan2 */
float orc_fast_atan2(float y, float x)
{
    const float scale = (float)(180.0 / 3.14159265358979323846);
    const float p1 = 0.9997878412794807f * scale;
    const float p3 = -0.3258083974640975f * scale;
    const float p5 = 0.1555786518463281f * scale;
    const float p7 = -0.04432655554792128f * scale;
    float ax = fabsf(x), ay = fabsf(y);
    float a, c, c2;
    if (ax >= ay) {
        c = ay / (ax + (float)DBL_EPSILON);
        c2 = c * c;
        a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    } else {
        c = ax / (ay + (float)DBL_EPSILON);
        c2 = c * c;
        a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
    }
    if (x < 0) a = 180.f - a;
    if (y < 0) a = 360.f - a;
    return a;
}

/* ------------------------------------------------------------------ A6: IC_Angle
 * ORBextractor.cpp:68-95 (centre already integral here) */
float orc_ic_angle(const uint8_t* img, int stride, int cx, int cy, const int32_t* umax)
{
    int m01 = 0, m10 = 0;
    const uint8_t* c = img + (size_t)cy * stride + cx;
    for (int u = -HALF_PATCH; u <= HALF_PATCH; ++u) m10 += u * c[u];
    for (int v = 1; v <= HALF_PATCH; ++v) {
        int vsum = 0, d = umax[v];
        for (int u = -d; u <= d; ++u) {
            int p = c[u + v * stride], m = c[u - v * stride];
            vsum += p - m;
            m10 += u * (p + m);
        }
        m01 += v * vsum;
    }
    return orc_fast_atan2((float)m01, (float)m10);
}

/* ------------------------------------------------------------------ A8: computeOrbDescriptor
 * ORBextractor.cpp:100-316.  Byte i uses pattern points 16i..16i+15; bit k of
 * byte i is I(point 16i+2k) < I(point 16i+2k+1). */
/* 0 (default): the shared fixed-order sine/cosine of include/ccm_sincos.h, which the HIP kernel evaluates too.
 * 1: this machine's libm cosf/sinf, i.e. literally what ORBextractor.cpp:105 calls.  The switch exists to MEASURE how far
 * the shared function departs from the reference's libm (tests/test_sincos_cpu.py); parity tests use 0. */
static int g_sincos_libm = 0;
void orc_set_sincos_libm(int on) { g_sincos_libm = on; }

void orc_orb_descriptor(const uint8_t* img, int stride, int cx, int cy, float angle_deg, uint8_t* desc)
{
    const float factorPI = (float)(3.14159265358979323846 / 180.f);   /* :98 */
    const float angle = angle_deg * factorPI;                          /* :104 */
    float a, b;
    if (g_sincos_libm) { a = cosf(angle); b = sinf(angle); }           /* :105 exactly as written */
    else ccm_sincosf(angle, &b, &a);                                   /* :105 a=cos, b=sin */
    const uint8_t* c = img + (size_t)cy * stride + cx;
    for (int i = 0; i < 32; i++) {
        int val = 0;
        for (int k = 0; k < 8; k++) {
            const signed char* pp = ccm_orb_pattern + 4 * (8 * i + k);
            const float x0 = pp[0], y0 = pp[1], x1 = pp[2], y1 = pp[3];
            const int t0 = c[rnd_f(x0 * b + y0 * a) * stride + rnd_f(x0 * a - y0 * b)];
            const int t1 = c[rnd_f(x1 * b + y1 * a) * stride + rnd_f(x1 * a - y1 * b)];
            val |= (t0 < t1) << k;
        }
        desc[i] = (uint8_t)val;
    }
}

/* ------------------------------------------------------------------ A5: DistributeOctTree
 * ORBextractor.cpp:707-931 with DivideNode :650-705.  The reference keeps the
 * nodes in a std::list and inserts children with push_front; here the list is
 * an index-linked pool.  One rule is OURS: the reference sorts (size, node
 * address) pairs (:852), so equal sizes are ordered by malloc addresses, which
 * is not reproducible; we order equal sizes by creation sequence (a node made
 * later counts as the larger address). */
typedef struct {
    int ulx, uly, urx, ury, blx, bly, brx, bry;
    int first, count;      /* slice of the key arena */
    int no_more;
    int prev, next;        /* list links */
    int seq;
} qnode;

typedef struct {
    qnode* nd; int nnd, cap;
    int* arena; int arena_n, arena_cap;
    int head, tail, size;
    const int32_t* xy;
} qtree;

static int qt_new(qtree* t)
{
    if (t->nnd == t->cap) { t->cap *= 2; t->nd = (qnode*)realloc(t->nd, sizeof(qnode) * t->cap); }
    int i = t->nnd++;
    memset(&t->nd[i], 0, sizeof(qnode));
    t->nd[i].seq = i; t->nd[i].prev = t->nd[i].next = -1;
    return i;
}
static int* qt_alloc_keys(qtree* t, int n, int* first)
{
    if (t->arena_n + n > t->arena_cap) {
        while (t->arena_n + n > t->arena_cap) t->arena_cap *= 2;
        t->arena = (int*)realloc(t->arena, sizeof(int) * t->arena_cap);
    }
    *first = t->arena_n; t->arena_n += n;
    return t->arena + *first;
}
static void qt_push_front(qtree* t, int i)
{
    t->nd[i].prev = -1; t->nd[i].next = t->head;
    if (t->head >= 0) t->nd[t->head].prev = i; else t->tail = i;
    t->head = i; t->size++;
}
static void qt_push_back(qtree* t, int i)
{
    t->nd[i].next = -1; t->nd[i].prev = t->tail;
    if (t->tail >= 0) t->nd[t->tail].next = i; else t->head = i;
    t->tail = i; t->size++;
}
static int qt_erase(qtree* t, int i)   /* returns next */
{
    int p = t->nd[i].prev, n = t->nd[i].next;
    if (p >= 0) t->nd[p].next = n; else t->head = n;
    if (n >= 0) t->nd[n].prev = p; else t->tail = p;
    t->size--;
    return n;
}

/* DivideNode: returns child ids in c[0..3] (n1..n4), -1 never (children always made) */
static void qt_divide(qtree* t, int pi, int c[4])
{
    qnode P = t->nd[pi];
    const int halfX = (int)ceilf((float)(P.urx - P.ulx) / 2);
    const int halfY = (int)ceilf((float)(P.bry - P.uly) / 2);
    for (int k = 0; k < 4; k++) c[k] = qt_new(t);
    qnode* n1 = &t->nd[c[0]]; qnode* n2 = &t->nd[c[1]]; qnode* n3 = &t->nd[c[2]]; qnode* n4 = &t->nd[c[3]];
    n1->ulx = P.ulx; n1->uly = P.uly; n1->urx = P.ulx + halfX; n1->ury = P.uly;
    n1->blx = P.ulx; n1->bly = P.uly + halfY; n1->brx = P.ulx + halfX; n1->bry = P.uly + halfY;
    n2->ulx = n1->urx; n2->uly = n1->ury; n2->urx = P.urx; n2->ury = P.ury;
    n2->blx = n1->brx; n2->bly = n1->bry; n2->brx = P.urx; n2->bry = P.uly + halfY;
    n3->ulx = n1->blx; n3->uly = n1->bly; n3->urx = n1->brx; n3->ury = n1->bry;
    n3->blx = P.blx; n3->bly = P.bly; n3->brx = n1->brx; n3->bry = P.bly;
    n4->ulx = n3->urx; n4->uly = n3->ury; n4->urx = n2->brx; n4->ury = n2->bry;
    n4->blx = n3->brx; n4->bly = n3->bry; n4->brx = P.brx; n4->bry = P.bry;
    int cnt[4] = { 0, 0, 0, 0 };
    int* which = (int*)malloc(sizeof(int) * (P.count > 0 ? P.count : 1));
    const int splitx = t->nd[c[0]].urx, splity = t->nd[c[0]].bry;
    for (int i = 0; i < P.count; i++) {
        const int key = t->arena[P.first + i];
        const float kx = (float)t->xy[2 * key], ky = (float)t->xy[2 * key + 1];
        int q;
        if (kx < splitx) q = (ky < splity) ? 0 : 2;
        else q = (ky < splity) ? 1 : 3;
        which[i] = q; cnt[q]++;
    }
    int firsts[4];
    for (int q = 0; q < 4; q++) {
        qt_alloc_keys(t, cnt[q], &firsts[q]);
        t->nd[c[q]].first = firsts[q];
        t->nd[c[q]].count = 0;
    }
    for (int i = 0; i < P.count; i++) {
        qnode* ch = &t->nd[c[which[i]]];
        t->arena[ch->first + ch->count++] = t->arena[t->nd[pi].first + i];
    }
    for (int q = 0; q < 4; q++) if (t->nd[c[q]].count == 1) t->nd[c[q]].no_more = 1;
    free(which);
}

typedef struct { int size, seq, node; } qpair;
static int qpair_cmp(const void* a, const void* b)
{
    const qpair* x = (const qpair*)a; const qpair* y = (const qpair*)b;
    if (x->size != y->size) return x->size < y->size ? -1 : 1;
    return x->seq < y->seq ? -1 : (x->seq > y->seq);
}

int orc_distribute_octree(const int32_t* xy, const int32_t* score, int n, int minX, int maxX,
                          int minY, int maxY, int N, int32_t* out_idx)
{
    if (n <= 0) return 0;
    qtree t;
    t.cap = 64; t.nnd = 0; t.nd = (qnode*)malloc(sizeof(qnode) * t.cap);
    t.arena_cap = 4 * n + 64; t.arena_n = 0; t.arena = (int*)malloc(sizeof(int) * t.arena_cap);
    t.head = t.tail = -1; t.size = 0; t.xy = xy;

    const int nIni = (int)roundf((float)(maxX - minX) / (maxY - minY));   /* :711 */
    const float hX = (float)(maxX - minX) / nIni;                          /* :713 */
    if (nIni < 1) { free(t.nd); free(t.arena); return -1; }
    int* root = (int*)malloc(sizeof(int) * nIni);
    int* rcount = (int*)calloc(nIni, sizeof(int));
    int* rwhich = (int*)malloc(sizeof(int) * n);
    for (int i = 0; i < n; i++) {
        int r = (int)((float)xy[2 * i] / hX);                             /* :737 */
        if (r >= nIni) r = nIni - 1;
        rwhich[i] = r; rcount[r]++;
    }
    for (int i = 0; i < nIni; i++) {
        int id = qt_new(&t);
        qnode* q = &t.nd[id];
        q->ulx = (int)(hX * (float)i); q->uly = 0;                        /* :723-726 */
        q->urx = (int)(hX * (float)(i + 1)); q->ury = 0;
        q->blx = q->ulx; q->bly = maxY - minY;
        q->brx = q->urx; q->bry = maxY - minY;
        qt_alloc_keys(&t, rcount[i], &q->first);
        q->count = 0;
        qt_push_back(&t, id);
        root[i] = id;
    }
    for (int i = 0; i < n; i++) { qnode* q = &t.nd[root[rwhich[i]]]; t.arena[q->first + q->count++] = i; }
    for (int it = t.head; it >= 0;) {                                      /* :742-753 */
        if (t.nd[it].count == 1) { t.nd[it].no_more = 1; it = t.nd[it].next; }
        else if (t.nd[it].count == 0) it = qt_erase(&t, it);
        else it = t.nd[it].next;
    }
    free(root); free(rcount); free(rwhich);

    int finish = 0;
    qpair* expand = (qpair*)malloc(sizeof(qpair) * (size_t)(4 * n + 16));
    qpair* prevexp = (qpair*)malloc(sizeof(qpair) * (size_t)(4 * n + 16));
    int nexp = 0;
    while (!finish) {
        int prevSize = t.size;
        int nToExpand = 0;
        nexp = 0;
        for (int it = t.head; it >= 0;) {                                  /* :774-833 */
            if (t.nd[it].no_more) { it = t.nd[it].next; continue; }
            int c[4];
            qt_divide(&t, it, c);
            for (int q = 0; q < 4; q++) {
                if (t.nd[c[q]].count > 0) {
                    qt_push_front(&t, c[q]);
                    if (t.nd[c[q]].count > 1) {
                        nToExpand++;
                        expand[nexp].size = t.nd[c[q]].count; expand[nexp].seq = t.nd[c[q]].seq;
                        expand[nexp].node = c[q]; nexp++;
                    }
                }
            }
            it = qt_erase(&t, it);
        }
        if (t.size >= N || t.size == prevSize) finish = 1;                 /* :837 */
        else if (t.size + nToExpand * 3 > N) {                             /* :841 */
            while (!finish) {
                prevSize = t.size;
                int nprev = nexp;
                memcpy(prevexp, expand, sizeof(qpair) * nprev);
                nexp = 0;
                qsort(prevexp, nprev, sizeof(qpair), qpair_cmp);          /* :852 (+ our tie rule) */
                for (int j = nprev - 1; j >= 0; j--) {
                    int c[4];
                    qt_divide(&t, prevexp[j].node, c);
                    for (int q = 0; q < 4; q++) {
                        if (t.nd[c[q]].count > 0) {
                            qt_push_front(&t, c[q]);
                            if (t.nd[c[q]].count > 1) {
                                expand[nexp].size = t.nd[c[q]].count; expand[nexp].seq = t.nd[c[q]].seq;
                                expand[nexp].node = c[q]; nexp++;
                            }
                        }
                    }
                    qt_erase(&t, prevexp[j].node);
                    if (t.size >= N) break;                                 /* :898 */
                }
                if (t.size >= N || t.size == prevSize) finish = 1;         /* :902 */
            }
        }
    }
    int m = 0;
    for (int it = t.head; it >= 0; it = t.nd[it].next) {                   /* :912-928 */
        const qnode* q = &t.nd[it];
        int best = t.arena[q->first];
        float mr = (float)score[best];
        for (int k = 1; k < q->count; k++) {
            int key = t.arena[q->first + k];
            if ((float)score[key] > mr) { best = key; mr = (float)score[key]; }
        }
        out_idx[m++] = best;
    }
    free(expand); free(prevexp); free(t.nd); free(t.arena);
    return m;
}

/* ------------------------------------------------------------------ A3 + A9: operator()
 * ORBextractor.cpp:933-1024 and :1216-1278 */
int orc_orb_extract(const orc_orb_params* p, const uint8_t* img, int w, int h, int stride,
                    orc_keypoint* kps, uint8_t* desc, int max_kps,
                    uint8_t* level_out, size_t level_out_bytes,
                    int cand_level, int32_t* cand_xy, int32_t* cand_score, int cand_max, int32_t* cand_n)
{
    if (!p || !img || w <= 0 || h <= 0) return 0;           /* empty image: silent return (:1219) */
    float scale[16], inv[16];
    int32_t nfeat[16], umax[16], lw[16], lh[16];
    if (orc_orb_tables(p, scale, inv, 0, 0, nfeat, umax)) return -1;
    orc_orb_level_sizes(p, w, h, lw, lh);
    const int L = p->nlevels;
    uint8_t* lev[16];
    size_t off = 0;
    for (int l = 0; l < L; l++) {
        lev[l] = (uint8_t*)malloc((size_t)lw[l] * lh[l]);
        if (l == 0) for (int y = 0; y < h; y++) memcpy(lev[0] + (size_t)y * w, img + (size_t)y * stride, w);
        else orc_resize_linear_u8(lev[l - 1], lw[l - 1], lh[l - 1], lw[l - 1], lev[l], lw[l], lh[l], lw[l]); /* :1293 */
        if (level_out && off + (size_t)lw[l] * lh[l] <= level_out_bytes)
            memcpy(level_out + off, lev[l], (size_t)lw[l] * lh[l]);
        off += (size_t)lw[l] * lh[l];
    }
    if (cand_n) *cand_n = 0;
    int total = 0, rc = 0;
    const int capc = 1 << 20;
    int32_t* cxy = (int32_t*)malloc(sizeof(int32_t) * 2 * capc);
    int32_t* csc = (int32_t*)malloc(sizeof(int32_t) * capc);
    int32_t* sel = (int32_t*)malloc(sizeof(int32_t) * capc);
    int32_t fxy[2 * 4096], fsc[4096];
    for (int l = 0; l < L && rc == 0; l++) {
        const int minBX = EDGE_TH - 3, minBY = minBX;                     /* :941-944 */
        const int maxBX = lw[l] - EDGE_TH + 3, maxBY = lh[l] - EDGE_TH + 3;
        const float width = (float)(maxBX - minBX), height = (float)(maxBY - minBY);
        const int nCols = (int)(width / 30.f), nRows = (int)(height / 30.f);   /* :952-953 */
        int nc = 0;
        if (nCols >= 1 && nRows >= 1) {
            const int wCell = (int)ceilf(width / nCols), hCell = (int)ceilf(height / nRows);
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
                    const int x0 = (int)iniX, y0 = (int)iniY, cw = (int)maxX - x0, ch = (int)maxY - y0;
                    const uint8_t* sub = lev[l] + (size_t)y0 * lw[l] + x0;
                    int k = orc_fast9_16(sub, cw, ch, lw[l], p->ini_th, fxy, fsc, 4096);      /* :978 */
                    if (k == 0) k = orc_fast9_16(sub, cw, ch, lw[l], p->min_th, fxy, fsc, 4096); /* :983 */
                    for (int m = 0; m < k && nc < capc; m++) {
                        cxy[2 * nc] = fxy[2 * m] + j * wCell;                /* :991-992 */
                        cxy[2 * nc + 1] = fxy[2 * m + 1] + i * hCell;
                        csc[nc] = fsc[m];
                        nc++;
                    }
                }
            }
        }
        if (l == cand_level && cand_n) {
            *cand_n = nc;
            for (int m = 0; m < nc && m < cand_max; m++) {
                if (cand_xy) { cand_xy[2 * m] = cxy[2 * m]; cand_xy[2 * m + 1] = cxy[2 * m + 1]; }
                if (cand_score) cand_score[m] = csc[m];
            }
        }
        int ns = orc_distribute_octree(cxy, csc, nc, minBX, maxBX, minBY, maxBY, nfeat[l], sel); /* :1003 */
        if (ns < 0) { rc = -1; break; }
        if (ns == 0) continue;
        uint8_t* blurred = (uint8_t*)malloc((size_t)lw[l] * lh[l]);
        orc_blur7_sigma2(lev[l], lw[l], lh[l], lw[l], blurred, lw[l]);          /* :1258-1259 */
        const int scaledPatch = (int)(PATCH_SIZE * scale[l]);                   /* :1006 */
        for (int m = 0; m < ns; m++) {
            if (total >= max_kps) { rc = -4; break; }
            const int kx = cxy[2 * sel[m]] + minBX, ky = cxy[2 * sel[m] + 1] + minBY;  /* :1012-1013 */
            orc_keypoint* kp = &kps[total];
            kp->angle = orc_ic_angle(lev[l], lw[l], kx, ky, umax);              /* :1022 */
            kp->response = (float)csc[sel[m]];
            kp->octave = l; kp->class_id = -1;
            kp->size = (float)scaledPatch;
            orc_orb_descriptor(blurred, lw[l], kx, ky, kp->angle, desc + (size_t)32 * total); /* :1263 */
            kp->x = (float)kx; kp->y = (float)ky;
            if (l != 0) { kp->x *= scale[l]; kp->y *= scale[l]; }               /* :1268-1274 */
            total++;
        }
        free(blurred);
    }
    for (int l = 0; l < L; l++) free(lev[l]);
    free(cxy); free(csc); free(sel);
    return rc ? rc : total;
}

/* ------------------------------------------------------------------ all-core CPU baseline (bench.py)
 * The reference extracts and matches single-threaded per call; a server with many clients would run one call per core.  This
 * driver deals n frames (then the n - 1 consecutive pairs) to `threads` POSIX threads, each running the same scalar code as above,
 * and reports wall-clock seconds per phase.  Measurement scaffolding: no Python, no GIL, no per-frame allocation outside the
 * oracle's own. */
#include <pthread.h>
#include <time.h>
typedef struct {
    const orc_orb_params* par; const uint8_t* imgs; int n, w, h, threads, tid, max_kps;
    orc_keypoint* kps; uint8_t* desc; int32_t* counts; int phase;
} mt_job;
static void* mt_worker(void* arg)
{
    mt_job* j = (mt_job*)arg;
    if (j->phase == 0) {
        for (int f = j->tid; f < j->n; f += j->threads) {
            int32_t dummy = 0;
            j->counts[f] = orc_orb_extract(j->par, j->imgs + (size_t)f * j->w * j->h, j->w, j->h, j->w, j->kps + (size_t)f * j->max_kps,
                                           j->desc + (size_t)f * j->max_kps * 32, j->max_kps, 0, 0, -1, 0, 0, 0, &dummy);
        }
    } else {
        int32_t* bi = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)j->max_kps);
        for (int f = j->tid; f + 1 < j->n; f += j->threads)
            if (j->counts[f] > 0 && j->counts[f + 1] > 0)
                orc_hamming_match(j->desc + (size_t)f * j->max_kps * 32, j->counts[f], j->desc + (size_t)(f + 1) * j->max_kps * 32, j->counts[f + 1],
                                  bi, bi + j->max_kps, bi + 2 * j->max_kps);
        free(bi);
    }
    return 0;
}
static double mt_now(void) { struct timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
/* returns the total keypoint count (or < 0); seconds[0] = extraction, seconds[1] = matching */
long orc_bench_extract_match_mt(const orc_orb_params* par, const uint8_t* imgs, int n, int w, int h, int threads, double* seconds)
{
    if (!par || !imgs || n < 1 || threads < 1) return -1;
    const int max_kps = par->nfeatures + 4 * par->nlevels + 64;
    orc_keypoint* kps = (orc_keypoint*)malloc(sizeof(orc_keypoint) * (size_t)n * max_kps);
    uint8_t* desc = (uint8_t*)malloc((size_t)n * max_kps * 32);
    int32_t* counts = (int32_t*)calloc(n, sizeof(int32_t));
    pthread_t* th = (pthread_t*)malloc(sizeof(pthread_t) * threads);
    mt_job* jobs = (mt_job*)malloc(sizeof(mt_job) * threads);
    for (int phase = 0; phase < 2; phase++) {
        const double t0 = mt_now();
        for (int t = 0; t < threads; t++) {
            jobs[t] = (mt_job){ par, imgs, n, w, h, threads, t, max_kps, kps, desc, counts, phase };
            pthread_create(&th[t], 0, mt_worker, &jobs[t]);
        }
        for (int t = 0; t < threads; t++) pthread_join(th[t], 0);
        if (seconds) seconds[phase] = mt_now() - t0;
    }
    long total = 0;
    for (int f = 0; f < n; f++) total += counts[f] > 0 ? counts[f] : 0;
    free(kps); free(desc); free(counts); free(th); free(jobs);
    return total;
}

/* TEST SCAFFOLDING: runs the oracle's extractor, matcher and a small bundle adjustment under AddressSanitizer /
 * UndefinedBehaviourSanitizer (CPU build only: tests/test_oracle_asan_cpu.py compiles this file together with oracle/*.c).
 * The round-1 advisor asked whether computeOrbDescriptor's rotated pattern can read outside the blurred clone: it cannot
 * (tests/test_sincos_cpu.py proves the bound), and this run shows the restatement performs no out-of-bounds access on
 * rectangle frames, blocky noise and a flat image. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../../oracle/oracle.h"

static unsigned long long s_state = 0x9E3779B97F4A7C15ull;
static unsigned rnd(void) { s_state ^= s_state << 13; s_state ^= s_state >> 7; s_state ^= s_state << 17; return (unsigned)(s_state >> 32); }

static void make_frame(uint8_t* img, int w, int h, int kind)
{
    if (kind == 2) { memset(img, 90, (size_t)w * h); return; }                       /* flat */
    if (kind == 1) {                                                                  /* blocky noise: corners everywhere, also at the borders */
        for (int y = 0; y < h; y += 4) for (int x = 0; x < w; x += 4) {
            const uint8_t v = (uint8_t)(rnd() & 255);
            for (int dy = 0; dy < 4 && y + dy < h; dy++) for (int dx = 0; dx < 4 && x + dx < w; dx++) img[(size_t)(y + dy) * w + x + dx] = v;
        }
        return;
    }
    memset(img, 128, (size_t)w * h);
    for (int r = 0; r < 400; r++) {
        const int x0 = rnd() % w, y0 = rnd() % h, rw = 8 + rnd() % 89, rh = 8 + rnd() % 89; const uint8_t v = (uint8_t)(rnd() & 255);
        for (int y = y0; y < y0 + rh && y < h; y++) for (int x = x0; x < x0 + rw && x < w; x++) img[(size_t)y * w + x] = v;
    }
    for (size_t i = 0; i < (size_t)w * h; i++) { int v = img[i] + (int)(rnd() % 9) - 4; img[i] = (uint8_t)(v < 0 ? 0 : v > 255 ? 255 : v); }
}

int main(void)
{
    const orc_orb_params par = { 1000, 1.2f, 8, 20, 7 };
    const int sizes[3][2] = { {752, 480}, {321, 243}, {1241, 376} };
    const int cap = 4096;
    orc_keypoint* kps = (orc_keypoint*)malloc(sizeof(orc_keypoint) * cap);
    uint8_t* desc = (uint8_t*)malloc((size_t)cap * 32);
    uint8_t* prev = (uint8_t*)malloc((size_t)cap * 32);
    int nprev = 0, total = 0;
    for (int s = 0; s < 3; s++)
        for (int kind = 0; kind < 3; kind++) {
            const int w = sizes[s][0], h = sizes[s][1];
            uint8_t* img = (uint8_t*)malloc((size_t)w * h);          /* exact size: any read past it is an ASan error */
            make_frame(img, w, h, kind);
            int32_t cn = 0;
            const int n = orc_orb_extract(&par, img, w, h, w, kps, desc, cap, 0, 0, -1, 0, 0, 0, &cn);
            if (n < 0) { fprintf(stderr, "extract failed %d\n", n); return 1; }
            total += n;
            if (n > 0 && nprev > 0) {
                int32_t* bi = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)n);
                orc_hamming_match(desc, n, prev, nprev, bi, bi + n, bi + 2 * n);
                free(bi);
            }
            memcpy(prev, desc, (size_t)n * 32); nprev = n;
            free(img);
        }
    free(kps); free(desc); free(prev);
    printf("ok %d keypoints\n", total);
    return total > 3000 ? 0 : 2;
}

// orb_types.h -- device-visible geometry tables of the ORB pipeline.
#pragma once
#include <cstdint>

#define ORB_MAX_LEVELS 16
#define ORB_EDGE 19            // EDGE_THRESHOLD, cslam/src/ORBextractor.cpp:65
#define ORB_BORDER 16          // EDGE_THRESHOLD-3 = minBorderX/Y, :941
#define ORB_HALF_PATCH 15
#define ORB_MAX_ROOTS 8
#define ORB_PATCH_R 21         // 18 (max rotated pattern offset) + 3 (blur taps)
#define ORB_PATCH_D 43
#define ORB_BLUR_D 37

struct OrbLevel {
    // level pixels: img + frame*plane + y*pitch + x.  Level 0 points at the caller's images.
    const uint8_t* img; long long plane; int pitch;
    int w, h;
    // FAST score map, same indexing
    uint8_t* smap; long long splane; int spitch;
    // bilinear tables that make this level from level-1 (device pointers; unused for level 0)
    const int* xofs; const short* xab; const int* yofs; const short* yab;
    // FAST cells of this level inside OrbGeom::cells
    int cell_first, ncells;
    // score tiles (64x16) of this level inside the flattened tile index space
    int tile_first, tiles_x, tiles_y;
    // DistributeOctTree inputs
    int quota;                 // mnFeaturesPerLevel[level]
    int roots; float hx;       // nIni, hX (ORBextractor.cpp:711-713)
    int bw, bh;                // maxBorder-minBorder extents
    int key_first, key_cap;    // per-frame candidate scratch (ping-pong) range
    int out_first, out_cap;    // per-frame selected-keypoint slots
    float scale;               // mvScaleFactor[level]
    float kp_size;             // (int)(PATCH_SIZE*scale)
};

struct OrbCell {
    short level, x0, y0, cw, ch, pad;   // sub-image rectangle passed to FAST (:978)
    int slot_first, slot_cap;           // per-frame candidate slots of this cell
};

// One workgroup of the fused FAST kernel: a run of cells of one cell row (k_fast_cells)
struct OrbBand {
    int cell_first, ncells;            // consecutive cells of one row of one level
    short level, xa, y0, pitch, bh, pad; // tile origin (xa multiple of 4, one spare dword left of the first cell), LDS pitch, rows
};

struct OrbGeom {
    int nlevels, ncells, ntiles;
    int frame0;              // first frame of this launch: kernels index frame blockIdx + frame0 (chunked host-buffer pipeline)
    int ini_th, min_th;
    int slots_per_frame;     // sum of slot_cap
    int keys_per_frame;      // sum of key_cap
    int out_per_frame;       // sum of out_cap
    int list_cap;            // max(quota)+4: node-list capacity of the octree kernel
    int umax[16];
    OrbLevel lv[ORB_MAX_LEVELS];
};
static_assert(sizeof(OrbGeom) <= 3584, "OrbGeom is passed by value in the kernel arguments (4 KiB limit)");

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

// A band of k_fast_cells: consecutive cells of one cell row of one level, with everything the kernel needs before it can request the
// band's pixels -- one 64-byte scalar load.  (Until round 3 the kernel loaded a 16-byte band record, THEN the level record it
// pointed to, THEN the first and last cell records, and computed two integer reciprocals: 3,700 cycles of dependent scalar work in
// front of the pixel loads of a workgroup that lives 26,000.)
struct OrbBand {
    const uint8_t* img;                // level image base; nullptr for level 0, whose pointer, pitch and plane change per call (OrbGeom::lv[0])
    long long plane;                   // bytes per frame of the level image
    int lpitch, w, h;                  // level pitch and size
    int level;
    int cell_first, ncells;            // consecutive cells of one row of one level
    short xa, y0, pitch, bh;           // tile origin (xa multiple of 4, one spare dword left of the first cell), LDS pitch, rows
    short c_lo, c_hi, dw_lo, dw_hi;    // detection columns of the tile [c_lo, c_hi); dwords with a detection column and both neighbours
    unsigned pw_inv, pq_inv;           // floor(2^20 / (pitch / 4)) + 1, floor(2^20 / (pitch / 16)) + 1: i / PW and i / PQ by multiply-shift
    // its (at most ORB_BAND_CELLS) cells: first detection column of the tile and number of detection columns, candidate slots of the cell
    short clo[4], cwd[4];
    int slot_first[4], slot_cap[4];
    int pad_[4];
};
#define ORB_BAND_CELLS 4
static_assert(sizeof(OrbBand) == 128, "two s_load_dwordx16");

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

// ba_types.h -- device view of one bundle-adjustment problem (all pointers are device memory).
#pragma once
#include <cstdint>

struct BaDev {
    int P, L, E, nfree;            // poses (all), landmarks (this rank's), edges (this rank's), free poses
    double* poses;                 // [P][7] qx qy qz qw tx ty tz
    double* Rt;                    // [P][12] R row-major, then t
    const double* intr;            // [P][4] fx fy cx cy
    const int* free_of;            // [P] free index or -1
    const int* pose_of_free;       // [nfree]
    double* points;                // [L][3]
    // edges sorted by (landmark, pose)
    const int* edge_pose; const int* edge_point;
    const double* obs;             // [E][2]
    const double* info;            // [E]
    uint8_t* active;               // [E] level 0
    double* err;                   // [E][2] last computed error
    const int* pt_first;           // [L+1] CSR landmark -> edges
    const int* pose_first;         // [nfree+1] CSR free pose -> pose_edges
    const int* pose_edges;
    // system
    double* Hpp; double* bp;       // [nfree][36], [nfree][6]
    double* Hll; double* bl;       // [L][9], [L][3]
    double* Hpl;                   // [E][18] 6x3 row-major
    double* Dinv;                  // [L][9]
    double* Hs; double* bs;        // (6 nfree)^2 row-major, upper block triangle; 6 nfree
    double* x;                     // [6 nfree + 3 L]
    // per LM trial (ba_sparse.hip): Z = Hpl L^-T per edge (L L^T = Hll + lambda I), db = Dinv b_l per landmark, ce = Hpl db per edge
    double* Z; double* db; double* ce;
};

// coarse level of the PCG preconditioner (ba_sparse.hip); Aci == nullptr switches it off
struct PcgCoarse {
    double* Aci;                   // [nc][nc] inverse of the coarse matrix, nc = 7 per aggregate
    double* rc;                    // [blocks of k_pcg_update][4][7] block partials of the restricted residual P^T r
    double* yc;                    // [nc] coarse correction
    double* cpart;                 // per workgroup of k_pcg_coarse: share of (P^T r) . yc
    const double* svec;            // [nfree][3] keyframe translations at the start of the call (the scale columns of P)
    const double* cen;             // [aggregates][3] mean translation of an aggregate's own keyframes
};

// buffers of the pipelined PCG (ba_sparse.hip, k_ppcg_*)
struct PpcgBufs {
    double* Hf;                    // [row entries][36] the reduced matrix with both triangles, a row's blocks side by side
    int* ecol;                     // [row entries] block column
    double* CA;                    // [2 A][coarse pitch] per-keyframe contributions to the restricted vector, by position in the aggregate's support
};

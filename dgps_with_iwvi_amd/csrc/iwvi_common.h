// Shared declarations for the gfx950 IW-ELBO kernels (internal; the public ABI is include/iwvi_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/iwvi_hip.h"

namespace iwvi {

// thread-local error text behind iwvi_last_error()
void set_error(const char* fmt, ...);
int check_launch(const char* what);

static inline int round_up(int x, int m) { return (x + m - 1) / m * m; }

// ---- per-layer state layout (see include/iwvi_hip.h) ----------------------------------------
struct StateLayout {
    int Mp, nb;
    size_t off_Lm, off_Linv, off_LinvP, off_LrTP, off_QmuP, off_Zs, off_invls, off_kl, off_ws, bytes;
};
static inline size_t align256(size_t x) { return (x + 255) & ~size_t(255); }
static inline StateLayout state_layout(int M, int R) {
    StateLayout s;
    s.Mp = round_up(M, 32);
    s.nb = s.Mp / 32;
    size_t o = 0;
    s.off_Lm = o;    o = align256(o + sizeof(double) * s.Mp * s.Mp);
    s.off_Linv = o;  o = align256(o + sizeof(double) * s.Mp * s.Mp);
    s.off_LinvP = o; o = align256(o + sizeof(float) * s.nb * s.nb * 1024);
    s.off_LrTP = o;  o = align256(o + sizeof(float) * (size_t)R * s.nb * s.nb * 1024);
    s.off_QmuP = o;  o = align256(o + sizeof(float) * s.nb * 1024);
    s.off_Zs = o;    o = align256(o + sizeof(float) * s.Mp * 32);
    s.off_invls = o; o = align256(o + sizeof(float) * 32);
    s.off_kl = o;    o = align256(o + sizeof(double) * IWVI_MAX_R);
    {   // factorisation workspace: 16x16 blocks (17-double rows) of the lower triangle + inverses + scratch
        const size_t nbk = s.Mp / 16;
        const size_t blocks = nbk * (nbk + 1) / 2 + nbk + (nbk * nbk + 3) / 4;
        s.off_ws = o; o = align256(o + sizeof(double) * blocks * 16 * 17);
    }
    s.bytes = o;
    return s;
}

// MFMA-fragment packing of a [32*nbr x 32*nbk] matrix G for v_mfma_f32_32x32x2_f32:
// block (bi, bk) is 1024 floats; float4 number (q*64 + lane) of the block holds
//   G[32*bi + (lane & 31)][32*bk + 8*q + 4*(lane >> 5) + e],  e = 0..3,  q = 0..3
// so one global_load_dwordx4 per lane (1 KiB per wave, fully coalesced) feeds four MFMAs whose
// k-pairs are {8q+e, 8q+4+e} -- exactly the row pairs a 32x32 accumulator register holds, which
// lets an accumulator tile be re-used as the next MFMA's B operand (cdna guide section 3).
__host__ __device__ static inline size_t packed_index(int nbk, int bi, int bk, int q, int lane, int e) {
    return ((size_t)(bi * nbk + bk) * 4 + q) * 256 + lane * 4 + e;
}

}  // namespace iwvi

// Device code of libevac, part 5: the all-gather of the returned observation batch (BASELINE north_star; the sharded env,
// evacuation_amd/distributed.py) as PEER STORES over xGMI -- no library collective.
//
// Every rank owns a buffer gathered[world][rows][take] and has mapped every peer's buffer into its address space once
// (hipIpc).  One launch of k_peer_gather on rank r copies columns [0, take) of its local record slab [rows][row_words]
// (the rollout kernel's packed [obs | reward | terminated | truncated] records; take = obs_dim picks the observation) into
// slice r of EVERY rank's buffer: workgroups [p * W, (p + 1) * W) serve peer p, all peers at once -- on the fully connected
// xGMI mesh each of the seven links carries one slice.  The stores are plain coalesced dword stores (a wave writes 256
// contiguous bytes); the records' stride (row_words) is removed on the way, so no staging copy of the observation columns
// is needed.
//
// Why a kernel of its own and not RCCL's: the rollout workgroups hold every CU for the whole launch (one 1024-thread
// workgroup per CU, 448 of a SIMD's 512 vector registers), so a collective kernel that is to run UNDER the next rollout
// must fit in what is left -- 64 registers per lane.  This kernel is compiled for that (launch bounds 256, ~20 VGPRs, no
// LDS): its waves slot in beside the rollout's and the gather of chunk j - 1 overlaps the compute of chunk j by
// construction.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace evac {

constexpr int kMaxPeers = 16;
struct PeerPtrs {
    float* dst[kMaxPeers];
};

// TAKE > 0: compile-time column count (the division is a multiply); TAKE == 0: run-time `take`
template <int TAKE>
__global__ __launch_bounds__(256) void k_peer_gather(const float* __restrict__ src, unsigned n /* rows * take */, unsigned row_words,
                                                     unsigned take_rt, PeerPtrs peers, int my_rank, int world, unsigned slice_words,
                                                     int wgs_per_peer) {
    const unsigned take = TAKE > 0 ? (unsigned)TAKE : take_rt;
    // peer order rotated by the rank: at any moment the ranks write to different peers (no link is everybody's first)
    const int k = (int)blockIdx.x / wgs_per_peer, part = (int)blockIdx.x - k * wgs_per_peer;
    const int peer = (my_rank + 1 + k) % world;
    float* __restrict__ dst = peers.dst[peer] + (size_t)my_rank * slice_words;
    const unsigned stride = (unsigned)wgs_per_peer * 256u;
    unsigned i = (unsigned)part * 256u + threadIdx.x;
    // four independent loads in flight per lane, then four stores
    for (; i + 3u * stride < n; i += 4u * stride) {
        float v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned e = i + (unsigned)u * stride, r = e / take, c = e - r * take;
            v[u] = src[(size_t)r * row_words + c];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) dst[i + (unsigned)u * stride] = v[u];
    }
    for (; i < n; i += stride) {
        const unsigned r = i / take, c = i - r * take;
        dst[i] = src[(size_t)r * row_words + c];
    }
}

}  // namespace evac

// Tile binning for gfx950: inclusive scan of tiles_touched (a5), duplicateWithKeys (a6), the stable
// radix tile sort (a7) and tile ranges (a8).
//
// Rules: SURVEY.md 8(a).  All integer work -- results are bit-exact against the CPU restatement under oracle/.
//
// The published algorithm sorts R (tile<<32 | depth_bits) keys with one 64-bit radix sort.  The result of that
// stable sort is reproduced here with ~4x less HBM traffic by splitting it:
//   1. stable-sort the I = N*P instances by depth bits (32-bit keys, I << R) -- over the bits that VARY only, in digits of
//      up to nine bits: three passes for a scene spanning up to 2^4 in depth ("depth sort over the VARYING bits" below);
//   2. emit each instance's (tile, instance) pairs walking the instances in that depth order, so the pair stream
//      is already depth-ordered (ties in depth keep ascending instance index, exactly the order in which the
//      published duplicateWithKeys lays equal keys out);
//   3. stable-sort the pairs by tile id alone (13..16 bits => two 8-bit passes on 8-byte pairs).
// A stable sort by tile of a depth-ordered stream is the (tile, depth) order with ties by emission order, i.e.
// bit for bit the order of the 64-bit sort.  Tests compare point_list / ranges / reconstructed 64-bit keys with
// the oracle's single 64-bit stable sort.
//
// The radix passes are wave64 kernels.  One kernel per pass (radix_sweep_kernel): it ranks the keys of a block with
// 64-bit ballots (match-any over the digit bits), learns how many keys with each digit lie in earlier blocks by
// decoupled look-back over a status array, reorders the block through LDS and writes each digit's run contiguously;
// the digit totals of all passes are counted once up front (by the pair emission for the tile sort).  The pipeline's
// two sorts move (key, value) as ONE 8-byte element, so a digit run is written as one contiguous stretch instead of
// two half as long, and their last pass writes only what is read afterwards (the instance list; tile ids + instance
// list).  Element counts are read from device memory so the host never has to know R to launch (grids are sized by
// capacity).  (The older three-kernel passes -- histogram, row scan, scatter -- are profiles/r02_sort_three_kernel_passes.patch.)
#include "hs_common.h"

// Non-temporal hints of the large-frame pair emission (emit_pairs_kernel<true>), measured one by one at c4 on one box
// (preprocess + binning, ms; none: 2.005): 1 = the 8-byte rectangle gather 2.156 (worse: the gathers then miss the
// neighbours another lane just brought in), 2 = the 4-byte slot-start scatter into the render records 1.934, 4 = the
// streaming store of the pairs 1.953.  The two stores it is.
#ifndef HS_EMIT_NT_MASK
#define HS_EMIT_NT_MASK 6
#endif

namespace hs {

namespace {

// ---------------------------------------------------------------- scan (a5)
constexpr int kScanItems = 4;
constexpr int kScanTile = 256 * kScanItems;

__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t v, int lane) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        uint32_t t = __shfl_up(v, d);
        if (lane >= d) v += t;
    }
    return v;
}

// Block-wide inclusive scan of one value per thread (NW waves: 256 threads by default); returns the inclusive prefix and
// the block total through *total.
template <int NW = 4>
__device__ __forceinline__ uint32_t block_incl_scan(uint32_t v, uint32_t* s_wave /*[NW]*/, uint32_t* total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = wave_incl_scan(v, lane);
    if (lane == 63) s_wave[wave] = incl;
    __syncthreads();
    uint32_t add = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
        const uint32_t c = s_wave[w];
        if (w < wave) add += c;
        tot += c;
    }
    *total = tot;
    __syncthreads();
    return incl + add;
}

__global__ void __launch_bounds__(256) scan_reduce_kernel(const uint32_t* in, int64_t n, uint32_t* block_sums) {
    __shared__ uint32_t s_wave[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t v = 0;
    if (base + kScanItems <= n) {
        uint4 q = *reinterpret_cast<const uint4*>(in + base);
        v = q.x + q.y + q.z + q.w;
    } else {
        for (int i = 0; i < kScanItems; ++i)
            if (base + i < n) v += in[base + i];
    }
    uint32_t total;
    block_incl_scan(v, s_wave, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// Single block: exclusive scan of block_sums in place; writes the grand total to *total_out (may be null).  Four
// consecutive values per thread, so a cloud's few thousand block sums take a handful of barrier rounds.
__global__ void __launch_bounds__(256) scan_spine_kernel(uint32_t* block_sums, int nblocks, uint32_t* total_out) {
    __shared__ uint32_t s_wave[4];
    uint32_t carry = 0;
    for (int base = 0; base < nblocks; base += 1024) {
        const int i = base + threadIdx.x * 4;
        uint32_t v[4], sum = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) { v[k] = i + k < nblocks ? block_sums[i + k] : 0u; sum += v[k]; }
        uint32_t total;
        const uint32_t incl = block_incl_scan(sum, s_wave, &total);
        uint32_t run = carry + incl - sum;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            if (i + k < nblocks) block_sums[i + k] = run;
            run += v[k];
        }
        carry += total;
    }
    if (threadIdx.x == 0 && total_out) *total_out = carry;
}

__global__ void __launch_bounds__(256) scan_apply_kernel(const uint32_t* in, int64_t n, const uint32_t* block_sums,
                                                         uint32_t* out) {
    __shared__ uint32_t s_wave[4];
    const int64_t base = (int64_t)blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    uint32_t v[kScanItems];
    uint32_t sum = 0;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        v[i] = base + i < n ? in[base + i] : 0;
        sum += v[i];
    }
    uint32_t total;
    uint32_t incl = block_incl_scan(sum, s_wave, &total);
    uint32_t run = block_sums[blockIdx.x] + incl - sum;
#pragma unroll
    for (int i = 0; i < kScanItems; ++i) {
        run += v[i];
        if (base + i < n) out[base + i] = run;
    }
}

// Sums of 256 consecutive values (the pair emission's block size): with scan_spine_kernel, the block-exclusive pair
// offsets of the emission when they are computed ahead of it (frames of fewer than 2^21 instances).
__global__ void __launch_bounds__(256) scan_reduce256_kernel(const uint2* srect, int64_t n, uint32_t* block_sums) {
    __shared__ uint32_t s_wave[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t cnt = 0;
    if (i < n) { const uint2 rc = srect[i]; cnt = (rc.y & 0xFFFFu) * (rc.y >> 16); }
    uint32_t total;
    block_incl_scan(cnt, s_wave, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

int scan_u32(const uint32_t* in, int64_t n, uint32_t* spine, uint32_t* out, uint32_t* total_out, hipStream_t s) {
    const int nblk = ceil_div(n, kScanTile);
    scan_reduce_kernel<<<nblk, 256, 0, s>>>(in, n, spine);
    scan_spine_kernel<<<1, 256, 0, s>>>(spine, nblk, total_out);
    scan_apply_kernel<<<nblk, 256, 0, s>>>(in, n, spine, out);
    HS_LAUNCH_CHECK();
    return HS_OK;
}

// Start of the binning stage: the instance count for the depth sort, and cleared tile ranges (tiles without pairs
// must read (0,0)).
__global__ void __launch_bounds__(256) bin_prepare_kernel(hs_counters* c, uint32_t n_inst, uint2* ranges, int64_t ntiles,
                                                          uint32_t* zero, int64_t n_zero, uint32_t* zero2, int64_t n_zero2) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (int64_t t = i; t < n_zero; t += (int64_t)gridDim.x * 256) zero[t] = 0u;     // scratch of the depth sort
    for (int64_t t = i; t < n_zero2; t += (int64_t)gridDim.x * 256) zero2[t] = 0u;   // emission scan status + pair-sort scratch
    if (i < ntiles) ranges[i] = make_uint2(0u, 0u);
    if (i == 0) { c->overflow = 0u; c->reserved[1] = n_inst; c->reserved[2] = 0u; c->reserved[3] = 0u; c->reserved[4] = 0u; }
}

// ---------------------------------------------------------------- radix sort passes (a7)
// number of set bits of `m` below `lane`
__device__ __forceinline__ int mask_rank(uint64_t m, int lane) {
    return (int)__popcll(m & ((1ull << lane) - 1ull));
}

template <typename K>
__device__ __forceinline__ uint32_t digit_of(K k, int shift, uint32_t mask) {
    return (uint32_t)(k >> shift) & mask;
}

// One kernel per pass instead of histogram + scan + scatter: the digit totals of ALL passes are counted once, up front
// (they do not depend on the order of the keys), and a block learns how many keys with its digit lie in EARLIER blocks
// by decoupled look-back over a status array -- every block publishes the per-digit counts of its keys as soon as
// it has ranked them (flag 1 = "my count"), walks back over its predecessors adding their words until it meets one
// flagged 2 = "inclusive prefix up to and including me", then publishes its own inclusive prefix.  Count and flag share
// one 32-bit word (2 + 30 bits: hs_plan rejects sorts of 2^30 or more elements), written and read with agent-scope
// atomics, so no ordering between separate words is needed (and no fence: a __threadfence() writes back the XCD's
// whole L2).
// Progress.  Chain position = blockIdx.  Alone on the GPU that is safe: every XCD's dispatcher hands out its share of a
// 1-D grid in increasing blockIdx order, so the lowest unfinished block is always running.  It is NOT safe next to other
// kernels (seen with several processes on one GPU): the XCDs advance independently, so A's resident blocks may wait for an
// A-block whose XCD is full of B's blocks, which wait the same way for a B-block behind A's.  Hence:
//   * HELPING: a wave still waiting for a predecessor's words after kHelpAfter polls counts that block's digits itself
//     (they depend on the pass' input alone), publishes them by compare-and-swap and walks on (see the look-back): no wait
//     depends on a block that has not started, the chain advances under any dispatch order.  Helps are counted in
//     fail_word[5] (hs_counters.reserved[4]); the host takes them as the sign of a shared GPU and moves to ticket order;
//   * HS_SORT_TICKETS=1 in the environment (or hs_sort_tickets(1)) selects the TICKET instantiation, in which a block's
//     chain position is a ticket drawn from a per-pass counter when it STARTS (one atomic per block): position p < b then
//     means block p is already running, nobody ever waits for an unstarted block, nothing needs help -- the faster mode on
//     a shared GPU.  Alone it costs +30 us per frame at c3 (six passes of one same-address atomic per block), which is
//     why it is not the default;
//   * every wait is bounded all the same (kSpinLimit / kSpinLimitBlockIdx polls): status words some stray write damaged
//     end in `fail_word` = 2 (hs_counters.overflow for the pipeline: the frame renders empty, the blocks that start after
//     the verdict only release their successors, the host raises) instead of a hung GPU.
// Measured alternatives (c3 tile sort, us per pass; the three-kernel pass: 76): this walk with 8 words in flight 42,
// with 16 / 32 in flight 48 / 55; group sums (one word per 32 blocks and digit, filled by returning atomics, plus a
// member counter) instead of inclusive prefixes 85.
constexpr uint32_t kStAgg = 1u << 30, kStIncl = 2u << 30, kStMask = (1u << 30) - 1u;
// Polls per awaited word (each ~1-2 us: a sleep and an uncached load) before a wait gives up.  After kHelpAfter polls a
// waiting wave of a blockIdx-ordered pass has published the awaited words itself, and in ticket order a predecessor is
// running by construction: running into these bounds means damaged status words, not a slow neighbour.  The blockIdx
// bound is the shorter one because the cure the host applies (ticket order, step repeated) makes waiting longer pointless.
constexpr int kSpinLimit = 1 << 20, kSpinLimitBlockIdx = 1 << 14;

__device__ __forceinline__ void st_publish(uint32_t* p, uint32_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint32_t st_read(const uint32_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// The digit totals are kept in kGhistCopies copies (a block adds to copy blockIdx % kGhistCopies, readers sum them): a few
// thousand blocks adding to the same 256 words one after the other is a serial chain of same-address atomics.
// (kGhistCopies, kGhistWords, kTicketWords: hs_common.h)

// Digit totals of every pass: ghist[pass * 256 + digit].  4096 elements per workgroup.  PACKED: the keys are the .x of
// (key, value) pairs.
constexpr int kHistThreads = 1024, kHistTile = 4096;
template <typename K, bool PACKED>
__global__ void __launch_bounds__(kHistThreads) radix_ghist_kernel(const void* keys_v, const uint32_t* n_dev, int nbits,
                                                                   int passes, uint32_t* ghist) {
    __shared__ uint32_t s_hist[8 * 256];
    const int64_t n = *n_dev;
    const int64_t base = (int64_t)blockIdx.x * kHistTile;
    if (base >= n) return;
    for (int i = threadIdx.x; i < passes * 256; i += kHistThreads) s_hist[i] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kHistTile / kHistThreads; ++i) {
        const int64_t k = base + i * kHistThreads + threadIdx.x;
        const bool valid = k < n;
        K key = (K)0;
        if (valid) {
            if constexpr (PACKED) key = (K) reinterpret_cast<const uint2*>(keys_v)[k].x;
            else key = reinterpret_cast<const K*>(keys_v)[k];
        }
        const uint64_t vm = __ballot(valid);
        int pass = 0;
        for (int shift = 0, w = 0; shift < nbits; shift += w, ++pass) {
            w = (nbits - shift + (passes - pass) - 1) / (passes - pass);
            const uint32_t d = digit_of<K>(key, shift, (1u << w) - 1u);
            // (nearly) constant digits -- exponent bits of depth, high tile bits -- would serialise the wave's LDS
            // atomics on one bin: when the whole wave agrees, one lane adds the count
            const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
            if (__ballot(valid && d != d0) == 0ull) {
                if ((threadIdx.x & 63) == 0 && vm) atomicAdd(&s_hist[pass * 256 + d0], (uint32_t)__popcll(vm));
            } else if (valid) {
                atomicAdd(&s_hist[pass * 256 + d], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t* mine = ghist + (blockIdx.x % kGhistCopies) * (8 * 256);
    for (int i = threadIdx.x; i < passes * 256; i += kHistThreads) {
        const uint32_t c = s_hist[i];
        if (c) atomicAdd(&mine[i], c);
    }
}

// ---- depth sort over the VARYING bits only (the instances' 32-bit depth keys) ----
// Depths of a scene span a few octaves: z in [2, 10) gives 25 varying bits (23 of mantissa, 2 of exponent), so four 8-bit
// passes over all 32 bits sort seven bits that are the same in every key.  Whoever writes the depth keys (preprocess_fwd in
// a single-enqueue forward, depth_keys_kernel otherwise) ORs every visible key and its complement into kDepthBitsCopies
// copies of two words; a bit varies iff it is set in both.  The passes then sort bits [lo, hi) only, in ceil((hi - lo) / 9)
// digits of at most NINE bits (512-thread blocks, one bin per thread): three passes for any scene spanning up to 2^4 in
// depth, read from device memory by every kernel (the host never knows); four kernels are always enqueued, a pass with no
// digit left returns at once, and the last pass that has one writes the instance list.  Culled instances carry the
// all-ones key, are left out of the OR, and sort behind (or, when a visible key is all ones inside the window, among)
// the largest visible keys -- they have no pairs, so their place changes nothing downstream.
// (no arrays indexed by a run-time pass number: those would live in scratch memory)
struct DepthLayout {
    int npasses, lo, base, rem;   // digit p: width base + (p < rem), starting at lo + p * base + min(p, rem)
    __device__ __forceinline__ int width(int p) const { return base + (p < rem ? 1 : 0); }
    __device__ __forceinline__ int shift(int p) const { return lo + p * base + min(p, rem); }
};
__device__ __forceinline__ DepthLayout depth_layout_from(uint32_t o, uint32_t nz) {
    const uint32_t varying = o & nz;                       // some key has the bit set, some key has it clear
    DepthLayout L;
    L.lo = varying ? (int)__builtin_ctz(varying) : 0;
    const int hi = varying ? 32 - (int)__builtin_clz(varying) : 1;
    const int nbits = hi - L.lo;
    L.npasses = (nbits + kDepthDigitBits - 1) / kDepthDigitBits;
    L.base = nbits / L.npasses;
    L.rem = nbits - L.base * L.npasses;
    return L;
}
__device__ __forceinline__ DepthLayout depth_layout(const unsigned long long* bits, uint32_t tag) {
    uint32_t o = 0u, nz = 0u;
#pragma unroll
    for (int c = 0; c < kDepthBitsCopies; ++c) {   // (words of another frame -- another tag -- count as empty)
        const unsigned long long a = bits[2 * c], b = bits[2 * c + 1];
        o |= (uint32_t)(a >> 32) == tag ? (uint32_t)a : 0u;
        nz |= (uint32_t)(b >> 32) == tag ? (uint32_t)b : 0u;
    }
    return depth_layout_from(o, nz);
}

// Digit totals of the depth sort's passes (<= 4 digits of <= 9 bits, layout read from the device): ghist[pass * 512 + digit]
__global__ void __launch_bounds__(1024) depth_ghist_kernel(const uint2* pairs, const uint32_t* n_dev,
                                                           const unsigned long long* bits, uint32_t tag, uint32_t* ghist) {
    __shared__ uint32_t s_hist[4 * kDepthBins];
    const int64_t n = *n_dev;
    const int64_t base = (int64_t)blockIdx.x * 4096;
    if (base >= n) return;
    const DepthLayout L = depth_layout(bits, tag);
    for (int i = threadIdx.x; i < L.npasses * kDepthBins; i += 1024) s_hist[i] = 0;
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t k = base + i * 1024 + threadIdx.x;
        const bool valid = k < n;
        const uint32_t key = valid ? pairs[k].x : 0u;
        const uint64_t vm = __ballot(valid);
        for (int pass = 0; pass < L.npasses; ++pass) {
            const uint32_t d = (key >> L.shift(pass)) & ((1u << L.width(pass)) - 1u);
            // the top digit is nearly constant (a few exponent values): when the whole wave agrees, one lane adds the count
            const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
            if (__ballot(valid && d != d0) == 0ull) {
                if ((threadIdx.x & 63) == 0 && vm) atomicAdd(&s_hist[pass * kDepthBins + d0], (uint32_t)__popcll(vm));
            } else if (valid) {
                atomicAdd(&s_hist[pass * kDepthBins + d], 1u);
            }
        }
    }
    __syncthreads();
    uint32_t* mine = ghist + (blockIdx.x % kGhistCopies) * (8 * 256);
    for (int i = threadIdx.x; i < L.npasses * kDepthBins; i += 1024) {
        const uint32_t c = s_hist[i];
        if (c) atomicAdd(&mine[i], c);
    }
}

// One pass: rank the block's keys (stable: wave w owns ITEMS * 64 consecutive keys of the block, ITEMS rounds of 64
// consecutive keys), publish / look back, reorder through LDS, write each digit's run contiguously.
//   PACKED_IN : `in_keys` holds (key, value) uint2 elements (K = uint32_t), else K keys with the values in `in_vals`;
//   PACKED_OUT: `out_keys` receives uint2 elements, else keys go to `out_keys` (skipped when null) and values to
//               `out_vals`.
//   LOOK      : status words a thread requests at once during the look-back.
//   BLOCK     : threads = bins per block (256: digits of <= 8 bits; 512: <= 9 bits, the depth sort);
//   DYN       : the depth sort's dynamic form -- `dyn_bits` gives the digit layout (depth_layout), `dyn_pass` this launch's
//               pass; `in_keys` / `out_keys` are the two ping-pong buffers (pass p reads the first when p is even), the
//               last pass that has a digit writes the values to `out_vals`, a pass without one returns at once.
template <typename K, int ITEMS, int LOOK, bool PACKED_IN, bool PACKED_OUT, bool TICKET, int BLOCK = kSortBlock, bool DYN = false>
__global__ void __launch_bounds__(BLOCK) radix_sweep_kernel(const void* in_keys, const uint32_t* in_vals,
                                                            void* out_keys, uint32_t* out_vals, const uint32_t* n_dev,
                                                            int shift, uint32_t mask, uint32_t* status,
                                                            const uint32_t* ghist, uint32_t* ticket,
                                                            uint32_t* fail_word, uint32_t* kill_word,
                                                            const unsigned long long* dyn_bits = nullptr, int dyn_pass = 0,
                                                            uint32_t dyn_tag = 0u) {
    constexpr int TILE = ITEMS * BLOCK;
    constexpr int NW = BLOCK / 64;                 // waves per block
    constexpr int DBITS = BLOCK == 512 ? 9 : 8;   // bits of the widest digit
    static_assert(BLOCK == 256 || BLOCK == 512, "one bin per thread: 256 or 512 bins");
    __shared__ uint32_t s_cnt[NW][BLOCK];   // per-wave digit counters -> per-wave exclusive offsets
    __shared__ uint32_t s_dstart[BLOCK];    // block-local start of each digit's run
    __shared__ uint32_t s_gbase[BLOCK];     // global start of this block's run of each digit
    __shared__ K s_keys[TILE];
    __shared__ uint32_t s_vals[TILE];
    __shared__ uint32_t s_wave[NW];
    __shared__ uint32_t s_ticket, s_fail, s_n;
    bool dyn_last = false;
    if constexpr (DYN) {
        const DepthLayout L = depth_layout(dyn_bits, dyn_tag);
        if (dyn_pass >= L.npasses) return;          // nothing left to sort: the previous pass wrote the final list
        shift = L.shift(dyn_pass);
        mask = (mask & 0x80000000u) | ((1u << L.width(dyn_pass)) - 1u);
        dyn_last = dyn_pass == L.npasses - 1;
        if (dyn_pass & 1) { const void* t = in_keys; in_keys = out_keys; out_keys = const_cast<void*>(t); }
        ghist += dyn_pass * BLOCK;
    }

#ifdef HS_TESTING
    // libhdrsplat_test.so only (HS_FAULT_INJECT=late_block sets bit 31 of `mask`): block 1 starts ~3 ms late, as if its XCD
    // had no room for it -- the blocks behind it must help themselves (see the look-back below)
    if ((mask >> 31) != 0u) {
        mask &= 0x7FFFFFFFu;
        if (!TICKET && blockIdx.x == 1)
            for (int i = 0; i < 1000; ++i) __builtin_amdgcn_s_sleep(127);
    }
#endif
    if (threadIdx.x == 0) {
        s_ticket = TICKET ? atomicAdd(ticket, 1u) : blockIdx.x;
        // (one lane asks for the element count and the verdict of the passes so far -- both loads in flight together, and
        // the whole block takes the same way out)
        const uint32_t n0 = *n_dev;
        const uint32_t f0 = __hip_atomic_load(fail_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_n = n0;
        s_fail = f0 >= 2u ? 1u : 0u;
    }
#pragma unroll
    for (int w = 0; w < NW; ++w) s_cnt[w][threadIdx.x] = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    const int64_t n = s_n;
    const int bid = (int)s_ticket;         // this block's place in the look-back chain (see above)
    const int64_t base = (int64_t)bid * TILE;
    if (base >= n) return;
    const int cnt_block = (int)min((int64_t)TILE, n - base);
    uint32_t* const my_status = status + (int64_t)bid * BLOCK + threadIdx.x;
    // A pass that gave up left its output incomplete: the digit totals no longer describe what the later passes would read,
    // and a scatter by them could leave the buffer.  Blocks that start after the verdict (the rest of that pass, every
    // block of the passes behind it) only release their successors and go.
    if (s_fail) {
        st_publish(my_status, kStIncl);
        return;
    }

    uint32_t digit_base;   // keys of the whole array with a smaller digit
    {
        uint32_t tot = 0;
#pragma unroll
        for (int c = 0; c < kGhistCopies; ++c) tot += ghist[c * (8 * 256) + threadIdx.x];
        uint32_t all;
        digit_base = block_incl_scan<NW>(tot, s_wave, &all) - tot;
    }

    K key[ITEMS];
    uint32_t val[ITEMS];
    uint16_t rank[ITEMS];
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const int wbase = wave * (TILE / NW);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int loc = wbase + i * 64 + lane;
        const bool valid = loc < cnt_block;
        key[i] = (K) ~(K)0;
        val[i] = 0u;
        if (valid) {
            if constexpr (PACKED_IN) {
                const uint2 e = reinterpret_cast<const uint2*>(in_keys)[base + loc];
                key[i] = (K)e.x; val[i] = e.y;
            } else {
                key[i] = reinterpret_cast<const K*>(in_keys)[base + loc];
                val[i] = in_vals[base + loc];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int loc = wbase + i * 64 + lane;
        const bool valid = loc < cnt_block;
        const uint32_t d = digit_of<K>(key[i], shift, mask);
        uint64_t peers = __ballot(valid);   // match-any: lanes holding the same digit
#pragma unroll
        for (int b = 0; b < DBITS; ++b) {
            const uint64_t m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t before = s_cnt[wave][d];
        const uint32_t below = __popcll(peers & lt_mask);
        rank[i] = (uint16_t)(before + below);
        // the last peer publishes the new count (all peers read `before` first: same wave, in-order LDS)
        if (valid && (peers >> lane) == 1ull) s_cnt[wave][d] = before + below + 1;
    }
    __syncthreads();
    uint32_t my_tot;
    {   // per digit: the block's count goes out first, then the offsets over waves and the block-local run starts
        uint32_t c[NW];
        my_tot = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { c[w] = s_cnt[w][threadIdx.x]; my_tot += c[w]; }
        st_publish(my_status, (bid == 0 ? kStIncl : kStAgg) | my_tot);
        uint32_t run = 0;
#pragma unroll
        for (int w = 0; w < NW; ++w) { s_cnt[w][threadIdx.x] = run; run += c[w]; }
        uint32_t total;
        const uint32_t incl = block_incl_scan<NW>(my_tot, s_wave, &total);
        s_dstart[threadIdx.x] = incl - my_tot;
    }
    __syncthreads();
    // reorder through LDS while the predecessors' words arrive
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int loc = wbase + i * 64 + lane;
        if (loc < cnt_block) {
            const uint32_t d = digit_of<K>(key[i], shift, mask);
            const uint32_t pos = s_dstart[d] + s_cnt[wave][d] + rank[i];
            s_keys[pos] = key[i];
            s_vals[pos] = val[i];
        }
    }
    {   // look-back: keys with this thread's digit in earlier blocks (LOOK words requested at once: the walk is bound
        // by the latency of these uncached loads -- with hundreds of blocks ranking at the same time the nearest
        // inclusive prefix is far behind).
        // HELPING (blockIdx order): a block whose words are not there after kHelpAfter polls may simply not have STARTED
        // -- another process' blocks in its place, which may in turn wait for blocks of theirs that OUR waiting blocks keep
        // out.  Waiting cannot resolve that; but the silent block's digit counts are a function of the pass' input alone,
        // so the lanes still waiting for it count them themselves (one read of that block's 4096 keys, shared among
        // however many lanes of the wave are still here), offer them to everybody (compare-and-swap into the empty
        // words) and walk on.  No wait depends on an unstarted block any more: the chain advances under any dispatch
        // order, alone or next to other processes.
        uint32_t excl = 0;
        bool done = threadIdx.x > (mask & 0x7FFFFFFFu);   // a digit no key of this pass can have: nothing to look up
        bool failed = false;
        constexpr int kHelpAfter = 128;
        // (called by the lanes of the wave that are still polling block q -- all at the same poll count, so together)
        auto help = [&](int q) {
            const uint64_t here = __ballot(true);
            const int n_here = __popcll(here), my = mask_rank(here, lane);
            uint32_t* tab = s_cnt[wave];                       // (free since the reorder above; wave-private row)
            for (int i = my; i < 64; i += n_here) tab[i] = 0u;
            const int64_t qbase = (int64_t)q * TILE;           // a full tile: q < bid
            for (int i = my; i < TILE; i += n_here) {
                K k;
                if constexpr (PACKED_IN) k = (K) reinterpret_cast<const uint2*>(in_keys)[qbase + i].x;
                else k = reinterpret_cast<const K*>(in_keys)[qbase + i];
                const uint32_t d = digit_of<K>(k, shift, mask);
                if ((int)(d >> 6) == wave) atomicAdd(&tab[d & 63u], 1u);
            }
            const uint32_t c = tab[lane];                      // (same wave: LDS operations complete in order)
            atomicCAS(status + (int64_t)q * BLOCK + threadIdx.x, 0u, kStAgg | c);
            // tell the host (hs_counters.reserved[4], four words behind `overflow`): help was needed, i.e. the GPU is shared
            // with kernels that keep our blocks out -- ticket order, where nobody waits for an unstarted block, is the faster mode then
            if (my == 0) atomicAdd(fail_word + 5, 1u);
        };
        for (int p = bid - 1; p >= 0 && !done; p -= LOOK) {
            uint32_t v[LOOK];
#pragma unroll
            for (int j = 0; j < LOOK; ++j)
                v[j] = p - j >= 0 ? st_read(status + (int64_t)(p - j) * BLOCK + threadIdx.x) : kStIncl;
            // fast path: all LOOK words are published already (the usual case once the chain is moving) -- no loop, no
            // branch per word: add counts up to and including the first inclusive prefix
            bool all_ready = true;
#pragma unroll
            for (int j = 0; j < LOOK; ++j) all_ready = all_ready && (v[j] & ~kStMask) != 0u;
            if (all_ready) {
                bool alive = true;
#pragma unroll
                for (int j = 0; j < LOOK; ++j) {
                    excl += alive ? (v[j] & kStMask) : 0u;
                    alive = alive && (v[j] & ~kStMask) != kStIncl;
                }
                done = !alive;
                continue;
            }
#pragma unroll
            for (int j = 0; j < LOOK; ++j) {
                if (done) break;
                uint32_t x = v[j];
                // (bounded: see kSpinLimit; the loop must not be unrolled -- a known trip bound gets it
                // unrolled eightfold, 64 copies of the poll in this walk)
                int polls = 0;
#pragma clang loop unroll(disable)
                while ((x & ~kStMask) == 0u && polls < (TICKET ? kSpinLimit : kSpinLimitBlockIdx)) {
                    ++polls;
                    if constexpr (!TICKET) { if (polls == kHelpAfter) help(p - j); }
                    __builtin_amdgcn_s_sleep(1);
                    x = st_read(status + (int64_t)(p - j) * BLOCK + threadIdx.x);
                }
                if ((x & ~kStMask) == 0u) { failed = true; done = true; break; }
                excl += x & kStMask;
                done = (x & ~kStMask) == kStIncl;
            }
        }
        if (failed) {   // a predecessor's word never arrived: give up, tell the host, let the successors pass
            s_fail = 1u;
            atomicMax(fail_word, 2u);
            if (kill_word) *kill_word = 0u;
        }
        if (bid != 0) st_publish(my_status, kStIncl | ((excl + my_tot) & kStMask));
        s_gbase[threadIdx.x] = digit_base + excl;
    }
    __syncthreads();
    if (s_fail) return;
#pragma unroll 4
    for (int i = 0; i < ITEMS; ++i) {
        const int pos = i * BLOCK + threadIdx.x;
        if (pos < cnt_block) {
            const K k = s_keys[pos];
            const uint32_t d = digit_of<K>(k, shift, mask);
            const int64_t dst = (int64_t)s_gbase[d] + (pos - s_dstart[d]);
            if constexpr (DYN) {   // the depth sort: packed elements between the passes, the bare instance list at the end
                if (dyn_last) out_vals[dst] = s_vals[pos];
                else reinterpret_cast<uint2*>(out_keys)[dst] = make_uint2((uint32_t)k, s_vals[pos]);
            } else if constexpr (PACKED_OUT) {
                reinterpret_cast<uint2*>(out_keys)[dst] = make_uint2((uint32_t)k, s_vals[pos]);
            } else {
                if (out_keys) reinterpret_cast<K*>(out_keys)[dst] = k;
                out_vals[dst] = s_vals[pos];
            }
        }
    }
}

// Scratch of a sort (`tmp`: sort_tmp_bytes(n_launch)): digit totals, one ticket counter per pass, status words.
struct SortScratch {
    uint32_t* ghist; uint32_t* tickets; uint32_t* status; int64_t pass_words;
    SortScratch(void* tmp, int64_t n_launch, int tile, int bins = 256) {
        ghist = (uint32_t*)tmp;                       // [kGhistCopies][8][256]
        tickets = ghist + kGhistWords;                // [8] (+ the depth-bits words: hs_common.h)
        status = tickets + kTicketWords;              // [passes][sweep_pass_words(nblk, bins)]
        pass_words = sweep_pass_words((n_launch + tile - 1) / tile, bins);
    }
};

// Stable LSD sort of packed (u32 key, u32 value) elements on key bits [0, nbits): ping-pongs between the packed
// buffers p0 (input) and p1; the LAST pass writes the values to `vals_out` and, when `keys_out` is given, the keys to
// `keys_out` (its source is p0 when the pass count is odd, else p1 -- the caller makes sure the outputs do not overlay
// that buffer).  `zeroed`: an earlier kernel cleared sort_scratch_words(n_launch, passes, ITEMS * 256) words of `tmp`;
// `ghist_ready`: ... and the digit totals have been counted into it as well.
template <int ITEMS, int LOOK>
int radix_sort_packed(uint2* p0, uint2* p1, uint32_t* keys_out, uint32_t* vals_out, const uint32_t* n_dev, int64_t n_launch,
                      int nbits, void* tmp, uint32_t* fail_word, uint32_t* kill_word, hipStream_t s, bool zeroed,
                      bool ghist_ready) {
    if (n_launch <= 0) return HS_OK;
    constexpr int TILE = ITEMS * kSortBlock;
    const int nblk = ceil_div(n_launch, TILE);
    const int passes = sort_passes(nbits);
    const SortScratch sc(tmp, n_launch, TILE);
    if (!zeroed) HS_HIP_CHECK(hipMemsetAsync(tmp, 0, (size_t)sort_scratch_words(n_launch, passes, TILE) * 4, s));
    if (!ghist_ready)
        radix_ghist_kernel<uint32_t, true><<<ceil_div(n_launch, kHistTile), kHistThreads, 0, s>>>(p0, n_dev, nbits, passes, sc.ghist);
    uint2* in = p0; uint2* out = p1;
    int pass = 0;
    for (int shift = 0, w = 0; shift < nbits; shift += w, ++pass) {
        w = (nbits - shift + (passes - pass) - 1) / (passes - pass);
        uint32_t* st = sc.status + (int64_t)pass * sc.pass_words;
        const uint32_t mask = ((1u << w) - 1u) | (fault_injection() == 3 ? 0x80000000u : 0u);
        const uint32_t* gh = sc.ghist + 256 * pass;
        uint32_t* tk = sc.tickets + pass;
        const bool last = pass == passes - 1;
#define HS_SWEEP(PO, TK)                                                                                                 \
    radix_sweep_kernel<uint32_t, ITEMS, LOOK, true, PO, TK><<<nblk, kSortBlock, 0, s>>>(                                 \
        in, nullptr, last ? (void*)keys_out : (void*)out, last ? vals_out : nullptr, n_dev, shift, mask, st, gh, tk,  \
        fail_word, kill_word)
        if (sort_tickets()) { if (last) HS_SWEEP(false, true); else HS_SWEEP(true, true); }
        else { if (last) HS_SWEEP(false, false); else HS_SWEEP(true, false); }
#undef HS_SWEEP
        HS_LAUNCH_CHECK();
        uint2* t = in; in = out; out = t;
    }
    return HS_OK;
}

// ---------------------------------------------------------------- depth sort by counting (frames below 2^21 instances)
// Round 6.  The look-back passes above cost a frame of a million instances 63 us -- digit totals, three passes of ~18 us
// (each a chain of ~8 dependent round trips per block), one launch that finds nothing left to do -- for 8 MB of keys.  The
// same order (a stable sort has ONE result) from four launches without a chain between workgroups:
//   1. depth_msd_count_kernel: per block of depth_msd_tile(I) elements, the number of keys in each of the 4096 buckets of
//      the top kMsdBits VARYING key bits (the layout of those bits as the passes read it: DepthLayout) -- one row of u16;
//   2. depth_msd_colscan_kernel: per bucket the exclusive prefix of the rows down the column, and the column total;
//   3. depth_msd_scatter_kernel: block b re-reads its elements and writes each to start of its bucket + elements of the
//      bucket in earlier blocks + the slot an LDS atomic hands it: a counting sort by the top bits, stable from block to
//      block and in no particular order inside a (block, bucket) group -- the elements carry their instance numbers, which
//      ARE the input order, so step 4 can put equal keys right.  Culled instances (the all-ones key) are counted apart and
//      go straight to the END of the instance list, in index order (the passes leave them among the largest visible keys; they have no pairs, so where
//      they stand changes nothing downstream -- the order of the VISIBLE instances is the passes', bit for bit);
//   4. depth_range_sort_kernel: workgroup k takes the buckets that START inside positions [2048 k, 2048 (k + 1)) -- whole
//      buckets, so the range holds every key of its part of the key space -- and sorts them by (key, instance number) in
//      LDS: a distribution sort when the keys spread evenly over the range's span (no bucket of its 2048 above 16 members),
//      else stable least-significant-digit passes (the look-back passes' ranking, nothing published, nothing looked up)
//      over instance numbers and keys: the instance list -- and, on the way out, what the emission wants next to it: the
//      instances' tile rectangles in that order and the sums per 256 of them (GatherOut below).
//      A range of more than 4096 elements (a bucket of more than 2048: one 4096th of the key range holds that many
//      instances) does not fit: its workgroup then runs the same passes through memory, chunk by chunk -- correct, slow,
//      counted in hs_counters.reserved[2] so that the host can go back to the look-back passes (hs_depth_sort).
#ifndef HS_ABL_MSD
#define HS_ABL_MSD 0   // timing switch: 1 = the scatter stores nothing (wrong result)
#endif
__device__ __forceinline__ void hier_coarse_rect(uint2 rc, uint32_t& sx0, uint32_t& sy0, uint32_t& sw, uint32_t& sh);   // (below)
// What the kernel between the depth sort and the pair / element emission used to do in a launch of its own (gather_binfo_kernel,
// hier_gather_kernel: 15 us at c3): the tile rectangles in depth order and, per 256 consecutive instances of that order, the
// sum of their pair counts (and of their super-tile counts: the hierarchical form).  The range sort has every instance's
// final position in its hands: it does this on the way out.  srect == null: not asked for (the emission gathers by itself).
#ifndef HS_TUNE_FUSE_GATHER
#define HS_TUNE_FUSE_GATHER 1
#endif
struct GatherOut {
    const uint2* binfo; uint2* srect; uint32_t* bsum; uint32_t* bcsum; uint32_t n_all;
};
constexpr uint32_t kCulledKey = 0xFFFFFFFFu;   // depth key of a culled instance (no positive float has these bits)
struct MsdDigit { int shift; uint32_t mask; };
__device__ __forceinline__ MsdDigit msd_digit(const DepthLayout& L) {
    const int nbits = L.base * L.npasses + L.rem;
    const int w = min(kMsdBits, nbits);
    MsdDigit D;
    D.shift = L.lo + nbits - w;
    D.mask = (1u << w) - 1u;
    return D;
}

// Stable ranks of a wave's ITEMS rounds of 64 digits (round i: lane l holds element i * 64 + l of the wave's stretch):
// rank[i] = elements of the same digit in front of it in the wave's stretch; cnt[d] (the wave's own counter row, zero on
// entry) ends as the wave's count of digit d.  `valid`: bit i = this lane's round-i element exists (digits of the others: 0).
template <int ITEMS, int DBITS, typename CNT>
__device__ __forceinline__ void wave_rank(const uint32_t (&dig)[ITEMS], uint32_t valid, CNT* cnt, uint16_t (&rank)[ITEMS], int lane) {
    const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const bool on = (valid >> i) & 1u;
        const uint32_t d = dig[i];
        uint64_t peers = __ballot(on);   // match-any: lanes holding the same digit
        if (peers == 0ull) { rank[i] = 0; continue; }   // (a round behind the end of the data: uniform, nothing to rank)
#pragma unroll
        for (int b = 0; b < DBITS; ++b) {
            const uint64_t m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        const uint32_t before = cnt[d];
        const uint32_t below = __popcll(peers & lt_mask);
        rank[i] = (uint16_t)(before + below);
        // the last peer publishes the new count (all peers read `before` first: same wave, in-order LDS)
        if (on && (peers >> lane) == 1ull) cnt[d] = (CNT)(before + below + 1u);
    }
}

template <int ITEMS>   // (per 256 threads: tiles of 4096 run 1024 threads x 4, tiles of 1024 run 256 x 4)
__global__ void __launch_bounds__(1024) depth_msd_count_kernel(const uint2* pairs, const uint32_t* n_dev,
                                                              const unsigned long long* bits, uint32_t tag, uint32_t* counts2,
                                                              uint32_t* culled_rows) {
    __shared__ uint32_t s_hist[kMsdBuckets];
    __shared__ uint32_t s_culled;
    for (int t = threadIdx.x; t < kMsdBuckets; t += blockDim.x) s_hist[t] = 0u;
    if (threadIdx.x == 0) s_culled = 0u;
    const int64_t n = *n_dev;
    const int64_t base = (int64_t)blockIdx.x * (ITEMS * (int)blockDim.x);
    const MsdDigit D = msd_digit(depth_layout(bits, tag));
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int64_t k = base + i * (int)blockDim.x + threadIdx.x;
        const uint32_t key = k < n ? pairs[k].x : 0u;
        const bool culled = k < n && key == kCulledKey;
        const bool valid = k < n && !culled;
        // (neighbouring instances are not neighbours in depth: no point in looking for a wave-wide common bucket)
        if (valid) atomicAdd(&s_hist[(key >> D.shift) & D.mask], 1u);
        const uint64_t cm = __ballot(culled);
        if (cm && (threadIdx.x & 63) == 0) atomicAdd(&s_culled, (uint32_t)__popcll(cm));
    }
    __syncthreads();
    if (threadIdx.x == 0) culled_rows[blockIdx.x] = s_culled;
    uint32_t* row = counts2 + (int64_t)blockIdx.x * (kMsdBuckets / 2);   // u16 pairs: at most ITEMS * 256 = 4096 per bucket ... (*)
    for (int t = threadIdx.x; t < kMsdBuckets / 2; t += blockDim.x) {
        // (*) a block whose 4096 elements share ONE bucket would need 4096 = 0x1000: fits (u16 holds 65535)
        row[t] = s_hist[2 * t] | (s_hist[2 * t + 1] << 16);
    }
}

// 16 column pairs x 64 row slots per workgroup; a slot = up to 8 consecutive rows (<= 512 rows), kept in registers
__global__ void __launch_bounds__(1024) depth_msd_colscan_kernel(const uint32_t* counts2, int nrows, uint2* bases2, uint2* totals2) {
    __shared__ uint2 s_wsum[16][16];   // per wave (four row slots) and column pair
    const int c = threadIdx.x % 16, rs = threadIdx.x / 16;
    const int t2 = blockIdx.x * 16 + c;                                   // column pair: buckets 2 t2, 2 t2 + 1
    const int rps = (nrows + 63) / 64;                                    // consecutive rows per slot (<= 8)
    const int r0 = min(nrows, rs * rps), r1 = min(nrows, r0 + rps);
    uint32_t kept[8];
    uint2 sum = make_uint2(0u, 0u);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        kept[j] = r0 + j < r1 ? counts2[(int64_t)(r0 + j) * (kMsdBuckets / 2) + t2] : 0u;
        sum.x += kept[j] & 0xFFFFu; sum.y += kept[j] >> 16;
    }
    // rows above this thread's, same column pair: lanes are (row slot % 4, column pair) = (lane / 16, lane % 16)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint2 incl = sum;
    { const uint32_t ux = __shfl_up(incl.x, 16), uy = __shfl_up(incl.y, 16); if (lane >= 16) { incl.x += ux; incl.y += uy; } }
    { const uint32_t ux = __shfl_up(incl.x, 32), uy = __shfl_up(incl.y, 32); if (lane >= 32) { incl.x += ux; incl.y += uy; } }
    if (lane >= 48) s_wsum[wave][c] = incl;
    __syncthreads();
    uint2 run = make_uint2(incl.x - sum.x, incl.y - sum.y), total = make_uint2(0u, 0u);
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const uint2 v = s_wsum[w][c];
        if (w < wave) { run.x += v.x; run.y += v.y; }
        total.x += v.x; total.y += v.y;
    }
    if (rs == 0) totals2[t2] = total;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        if (r0 + j < r1) bases2[(int64_t)(r0 + j) * (kMsdBuckets / 2) + t2] = run;
        run.x += kept[j] & 0xFFFFu; run.y += kept[j] >> 16;
    }
}

// Exclusive scan of the kMsdBuckets totals into `s_start` (NT threads; `s_wave`: NT / 64 words): the bucket starts
template <int NT>
__device__ __forceinline__ uint32_t msd_bucket_starts(const uint32_t* totals, uint32_t* s_start, uint32_t* s_wave) {
    constexpr int E = kMsdBuckets / NT;   // consecutive buckets per thread
    for (int k = threadIdx.x; k < kMsdBuckets; k += NT) s_start[k] = totals[k];
    __syncthreads();
    uint32_t mine = 0;
#pragma unroll
    for (int j = 0; j < E; ++j) mine += s_start[threadIdx.x * E + j];
    uint32_t all;
    uint32_t start = block_incl_scan<NT / 64>(mine, s_wave, &all) - mine;   // (its barriers order the reads above before the writes below)
#pragma unroll
    for (int j = 0; j < E; ++j) { const uint32_t tot = s_start[threadIdx.x * E + j]; s_start[threadIdx.x * E + j] = start; start += tot; }
    __syncthreads();
    return all;   // = the visible instances
}

// NT threads x ITEMS elements: 512 x 8 for blocks of 4096, 256 x 4 for blocks of 1024.  The order in which the elements of one
// (block, bucket) group take their slots is whatever the LDS atomics make it: every element carries its instance number,
// instances are numbered in input order, and the range sort orders equal keys by that number -- the sort as a whole is the
// stable one, its first pass need not be.  (Ranking the group's members by match-any over the twelve bucket bits, one
// counter row per wave -- a stable pass -- took 6 of the kernel's 21 us at c3 plus the handling of 64 KB of counters.)
template <int NT, int ITEMS>
__global__ void __launch_bounds__(NT) depth_msd_scatter_kernel(const uint2* pairs, const uint32_t* n_dev,
                                                                const unsigned long long* bits, uint32_t tag,
                                                                const uint32_t* bases, const uint32_t* totals,
                                                                const uint32_t* culled_rows, uint2* out, uint32_t* inst_sorted,
                                                                uint32_t* zero_a, int64_t n_a, uint32_t* zero_b, int64_t n_b) {
    constexpr int NW = NT / 64;
    // (words the kernels behind the depth sort expect cleared: the block sums the range sort adds to, the hierarchical tile
    // sort's header and counters)
    for (int64_t t = (int64_t)blockIdx.x * NT + threadIdx.x; t < n_a; t += (int64_t)gridDim.x * NT) zero_a[t] = 0u;
    for (int64_t t = (int64_t)blockIdx.x * NT + threadIdx.x; t < n_b; t += (int64_t)gridDim.x * NT) zero_b[t] = 0u;
    __shared__ uint32_t s_cnt[kMsdBuckets];      // slots of the bucket this block has handed out
    __shared__ uint32_t s_base[kMsdBuckets];     // bucket start + elements of the bucket in earlier blocks
    __shared__ uint32_t s_wave[NW], s_cwave[NW], s_cbefore[NW];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int t = threadIdx.x; t < kMsdBuckets; t += NT) s_cnt[t] = 0u;
    const int64_t n = *n_dev;
    const int64_t base = (int64_t)blockIdx.x * (ITEMS * NT);
    const MsdDigit D = msd_digit(depth_layout(bits, tag));
    uint2 e[ITEMS];
    uint32_t valid = 0u, culled = 0u;
    const int wbase = wave * (ITEMS * 64);
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        const int64_t k = base + wbase + i * 64 + lane;
        e[i] = k < n ? pairs[k] : make_uint2(0u, 0u);
        const bool c = k < n && e[i].x == kCulledKey;
        valid |= ((k < n && !c) ? 1u : 0u) << i;
        culled |= (c ? 1u : 0u) << i;
    }
    const uint32_t* brow = bases + (int64_t)blockIdx.x * kMsdBuckets;
    uint32_t bv[kMsdBuckets / NT];
#pragma unroll
    for (int j = 0; j < kMsdBuckets / NT; ++j) bv[j] = brow[threadIdx.x + j * NT];
    // culled instances of the blocks in front of this one (<= 512 rows)
    uint32_t cb = 0;
    for (int r = threadIdx.x; r < (int)blockIdx.x; r += NT) cb += culled_rows[r];
#pragma unroll
    for (int dd = 32; dd >= 1; dd >>= 1) cb += __shfl_xor(cb, dd);
    if (lane == 0) s_cbefore[wave] = cb;
    const uint32_t n_vis = msd_bucket_starts<NT>(totals, s_base, s_wave);   // (its first barrier also covers the zeroing above)
#pragma unroll
    for (int j = 0; j < kMsdBuckets / NT; ++j) s_base[threadIdx.x + j * NT] += bv[j];
    // Culled instances (key all ones; none of them has a pair) take no part in the sort: they go behind the visible ones at
    // once, in index order -- a frame that sees a tenth of its cloud would otherwise carry the other nine tenths through the
    // range sort as ONE bucket of equal keys.
    uint32_t crank[ITEMS];
    uint32_t cwave = 0;
    {
        const uint64_t lt_mask = (1ull << lane) - 1ull;
#pragma unroll
        for (int i = 0; i < ITEMS; ++i) {
            const uint64_t m = __ballot((culled >> i) & 1u);
            crank[i] = cwave + (uint32_t)__popcll(m & lt_mask);
            cwave += (uint32_t)__popcll(m);
        }
        if (lane == 0) s_cwave[wave] = cwave;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < ITEMS; ++i) {
        if (!(HS_ABL_MSD & 1) && ((valid >> i) & 1u)) {
            const uint32_t d = (e[i].x >> D.shift) & D.mask;
            out[s_base[d] + atomicAdd(&s_cnt[d], 1u)] = e[i];
        }
    }
    if (culled) {
        uint32_t cfirst = n_vis;
#pragma unroll
        for (int w = 0; w < NW; ++w) cfirst += s_cbefore[w] + (w < wave ? s_cwave[w] : 0u);
#pragma unroll
        for (int i = 0; i < ITEMS; ++i)
            if ((culled >> i) & 1u) inst_sorted[cfirst + crank[i]] = e[i].y;
    }
}

constexpr int kRangeThreads = 512, kRangeItems = kMsdCap / kRangeThreads;   // 8 waves, 8 elements per thread
constexpr int kDistBits = 11, kDistBuckets = 1 << kDistBits;                 // distribution sort of a range: buckets of its relative keys' top bits
static_assert(kRangeThreads == kDepthBins, "one digit per thread: digits of up to nine bits");
__global__ void __launch_bounds__(kRangeThreads) depth_range_sort_kernel(const uint2* msd_sorted, uint2* scratch,
                                                                         const unsigned long long* bits, uint32_t tag,
                                                                         const uint32_t* totals, uint32_t* inst_sorted,
                                                                         hs_counters* counters, int cap, int ibits, int dist_max,
                                                                         GatherOut G, int range) {
    constexpr int NW = kRangeThreads / 64;
    // the passes: per-wave digit counters -> per-wave offsets [NW][512], start of each digit's run (or, off chip, its running
    // global start) [512]; the distribution sort: members per bucket, next free slot, first slot [3][kDistBuckets]
    __shared__ uint32_t s_work[3 * kDistBuckets];
    static_assert(3 * kDistBuckets >= (NW + 1) * kDepthBins, "the passes' tables fit the distribution sort's");
    uint32_t (*const s_cnt)[kDepthBins] = reinterpret_cast<uint32_t (*)[kDepthBins]>(s_work);
    uint32_t* const s_dstart = s_work + NW * kDepthBins;
    __shared__ uint32_t s_keys[kMsdCap], s_vals[kMsdCap], s_fin[kMsdCap];
    __shared__ uint32_t s_wave[NW];
    __shared__ uint32_t s_r[2], s_b[2];
    __shared__ uint32_t s_bs[64];   // pair counts [0, 32) and super-tile counts [32, 64) of the range's (up to 18) blocks of 256 positions
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < 2) { s_r[threadIdx.x] = 0xFFFFFFFFu; s_b[threadIdx.x] = (uint32_t)kMsdBuckets; }
    // this workgroup's range: [first bucket start >= 2048 k, first bucket start >= 2048 (k + 1)) (n_vis when there is none)
    const uint32_t n_vis = msd_bucket_starts<kRangeThreads>(totals, s_keys, s_wave);   // (the culled ones are in place already)
    {
        // (thresholds behind the last element mean "the end": the first bucket starting there, so that the last range's
        // span stops with its last occupied bucket instead of running on to bucket 4095)
        const uint32_t t0 = min(n_vis, (uint32_t)blockIdx.x * (uint32_t)range), t1 = min(n_vis, t0 + (uint32_t)range);
#pragma unroll
        for (int j = 0; j < kMsdBuckets / kRangeThreads; ++j) {
            const int b = threadIdx.x * (kMsdBuckets / kRangeThreads) + j;
            const uint32_t st = s_keys[b], prev = b ? s_keys[b - 1] : 0u;
            // (the first bucket at or behind a threshold: its predecessor starts in front of it -- or it is bucket 0)
            // (one bucket at most passes each test: plain stores)
            if (st >= t0 && (b == 0 || prev < t0)) { s_r[0] = st; s_b[0] = (uint32_t)b; }
            if (st >= t1 && (b == 0 || prev < t1)) { s_r[1] = st; s_b[1] = (uint32_t)b; }
        }
    }
    __syncthreads();
    const uint32_t r0 = min(s_r[0], n_vis), r1 = min(s_r[1], n_vis);
    if (G.srect) {   // the culled instances behind the visible ones have no rectangle: every workgroup clears a slice of that tail
        const uint32_t tail = G.n_all - n_vis, per = (tail + gridDim.x - 1) / gridDim.x;
        const uint32_t a = n_vis + min(tail, blockIdx.x * per), b = n_vis + min(tail, blockIdx.x * per + per);
        for (uint32_t t = a + threadIdx.x; t < b; t += kRangeThreads) G.srect[t] = make_uint2(0u, 0u);
    }
    if (r0 >= r1) return;
    // On the way out: the range's instances in their final order (`fin`, LDS) -> the instance list; with G.srect also their
    // rectangles and the block sums (see GatherOut)
    auto epilogue = [&](const uint32_t* fin, uint32_t n) {
        if (!G.srect) {
            for (uint32_t t = threadIdx.x; t < n; t += kRangeThreads) inst_sorted[r0 + t] = fin[t];
            return;
        }
        const bool ok = counters->overflow < 2u;
        const uint32_t g0 = r0 >> 8, slots = ((r0 + n - 1u) >> 8) - g0 + 1u;   // <= 18
        if (threadIdx.x < 64) s_bs[threadIdx.x] = 0u;
        __syncthreads();
        // (a wave's 64 consecutive positions lie in one block of 256 or two: it adds them up itself and touches the LDS
        // counters once or twice -- 512 threads adding to a dozen addresses would queue up behind each other)
        for (uint32_t t0 = 0; t0 < n; t0 += kRangeThreads) {
            const uint32_t t = t0 + threadIdx.x, pos = r0 + t;
            uint2 rc = make_uint2(0u, 0u);
            if (t < n) {
                const uint32_t inst = fin[t];
                inst_sorted[pos] = inst;
                if (ok) rc = G.binfo[inst];
                G.srect[pos] = rc;
            }
            uint32_t cnt = (rc.y & 0xFFFFu) * (rc.y >> 16), ccnt = 0;
            if (G.bcsum) {
                uint32_t sx0, sy0, sw, sh;
                hier_coarse_rect(rc, sx0, sy0, sw, sh);
                ccnt = sw * sh;
            }
            const uint32_t first_blk = __shfl(pos, 0) >> 8;
            const bool hi = (pos >> 8) != first_blk;
            uint32_t lo_c = hi ? 0u : cnt, hi_c = hi ? cnt : 0u, lo_cc = hi ? 0u : ccnt, hi_cc = hi ? ccnt : 0u;
#pragma unroll
            for (int dd = 32; dd >= 1; dd >>= 1) {
                lo_c += __shfl_xor(lo_c, dd); hi_c += __shfl_xor(hi_c, dd);
                lo_cc += __shfl_xor(lo_cc, dd); hi_cc += __shfl_xor(hi_cc, dd);
            }
            if (lane == 0) {
                const uint32_t sl = first_blk - g0;
                if (lo_c) atomicAdd(&s_bs[sl], lo_c);
                if (hi_c) atomicAdd(&s_bs[sl + 1u], hi_c);
                if (lo_cc) atomicAdd(&s_bs[32u + sl], lo_cc);
                if (hi_cc) atomicAdd(&s_bs[32u + sl + 1u], hi_cc);
            }
        }
        __syncthreads();
        if (threadIdx.x < slots && s_bs[threadIdx.x]) atomicAdd(&G.bsum[g0 + threadIdx.x], s_bs[threadIdx.x]);
        if (G.bcsum && threadIdx.x >= 32u && threadIdx.x < 32u + slots && s_bs[threadIdx.x])
            atomicAdd(&G.bcsum[g0 + threadIdx.x - 32u], s_bs[threadIdx.x]);
    };
    {   // the range's first OCCUPIED bucket: the last of the buckets starting at r0 (those in front of it are empty)
#pragma unroll
        for (int j = 0; j < kMsdBuckets / kRangeThreads; ++j) {
            const int b = threadIdx.x * (kMsdBuckets / kRangeThreads) + j;
            if (s_keys[b] == r0 && (b == kMsdBuckets - 1 || s_keys[b + 1] > r0)) s_b[0] = (uint32_t)b;
        }
        __syncthreads();
    }
    const uint32_t n = r1 - r0;
    const DepthLayout L0 = depth_layout(bits, tag);
    // The range's buckets [b0, b1) are in order already; what is left to sort are the key bits below the bucket's and the
    // bucket number RELATIVE to b0 -- a handful of bits for a range of a few buckets: sort (key window) - (b0's first key),
    // which has as many bits as (b1 - b0) << (bits below the bucket's) needs -- two digits instead of three at c3.
    const MsdDigit D = msd_digit(L0);
    const int nbits = L0.base * L0.npasses + L0.rem;
    const int low = D.shift - L0.lo;                                        // window bits below the bucket's
    const uint32_t b0 = s_b[0], b1 = s_b[1];
    const uint32_t wmask = nbits >= 32 ? 0xFFFFFFFFu : (1u << nbits) - 1u;
    const uint32_t kbase = b0 << low;                                       // window key of the range's first possible key
    const uint64_t span = (uint64_t)(b1 - b0) << low;                       // relative keys are < span
    DepthLayout L;
    {
        const int rbits = span > 1ull ? 64 - __builtin_clzll(span - 1ull) : 1;
        L.lo = 0;
        L.npasses = (rbits + kDepthDigitBits - 1) / kDepthDigitBits;
        L.base = rbits / L.npasses;
        L.rem = rbits - L.base * L.npasses;
    }
    DepthLayout IL;   // digits of the instance numbers (< 2^ibits), for the ranges that take passes
    IL.lo = 0;
    IL.npasses = (ibits + kDepthDigitBits - 1) / kDepthDigitBits;
    IL.base = ibits / IL.npasses;
    IL.rem = ibits - IL.base * IL.npasses;
    const int klo = L0.lo;
    auto rel_key = [&](uint32_t key) { return ((key >> klo) & wmask) - kbase; };
    // a wave owns ceil(n / 512) rounds of 64 consecutive elements (not a fixed eight: a range of 2048 keeps all eight waves
    // busy for four rounds each instead of four waves for eight)
    const int wbase = wave * (int)((min(n, (uint32_t)kMsdCap) + kRangeThreads - 1) / kRangeThreads) * 64;
    uint32_t key[kRangeItems], val[kRangeItems], dig[kRangeItems];
    uint16_t rank[kRangeItems];
    if (n <= (uint32_t)cap) {
        const uint32_t wend = min(n, (uint32_t)wbase + ((n + kRangeThreads - 1) / kRangeThreads) * 64u);   // end of this wave's stretch
        // ---- the range in LDS: the passes' ranking, digit after digit
        uint32_t valid = 0u;
#pragma unroll
        for (int i = 0; i < kRangeItems; ++i) {
            const uint32_t loc = wbase + i * 64 + lane;
            const bool on = loc < wend;
            const uint2 e = on ? msd_sorted[r0 + loc] : make_uint2(0u, 0u);
            key[i] = on ? rel_key(e.x) : 0u; val[i] = e.y;
            valid |= (on ? 1u : 0u) << i;
        }
        // Keys spread over the range's span as evenly as depths do: a distribution sort needs no pass at all.  2048 buckets
        // of the relative key's top eleven bits (LDS atomics: the order in which a bucket's members arrive is arbitrary), then
        // every element counts the members of ITS bucket that stand in front of it by (key, position in the input) -- a
        // handful of LDS reads where two or three ranking passes cost ~130 vector instructions per element.  A bucket of
        // more than `dist_max` (16) members (equal keys, a cluster) would make that quadratic: such a range takes the passes below.
        constexpr int E = kDistBuckets / kRangeThreads;   // consecutive buckets per thread
        const int rbits = L.base * L.npasses + L.rem;
        const int dsh = max(0, rbits - kDistBits);
        uint32_t* const hist = s_work;
        uint32_t* const cursor = s_work + kDistBuckets;
        uint32_t* const first = s_work + 2 * kDistBuckets;
#pragma unroll
        for (int j = 0; j < E; ++j) hist[threadIdx.x + j * kRangeThreads] = 0u;
        __syncthreads();
#pragma unroll
        for (int i = 0; i < kRangeItems; ++i)
            if ((valid >> i) & 1u) atomicAdd(&hist[key[i] >> dsh], 1u);
        __syncthreads();
        bool dist;
        {
            uint32_t c[E], mine = 0, most = 0;
#pragma unroll
            for (int j = 0; j < E; ++j) { c[j] = hist[threadIdx.x * E + j]; mine += c[j]; most = max(most, c[j]); }
            uint32_t total;
            uint32_t start = block_incl_scan<NW>(mine, s_wave, &total) - mine;
#pragma unroll
            for (int j = 0; j < E; ++j) { first[threadIdx.x * E + j] = start; cursor[threadIdx.x * E + j] = start; start += c[j]; }
            dist = __syncthreads_or(most > (uint32_t)dist_max) == 0;
        }
        if (dist) {
#pragma unroll
            for (int i = 0; i < kRangeItems; ++i) {
                if ((valid >> i) & 1u) {
                    const uint32_t p = atomicAdd(&cursor[key[i] >> dsh], 1u);
                    s_keys[p] = key[i];
                    s_vals[p] = val[i];
                }
            }
            __syncthreads();
            for (uint32_t p = threadIdx.x; p < n; p += kRangeThreads) {
                const uint32_t k = s_keys[p], me = s_vals[p];
                const uint32_t d = k >> dsh, a = first[d], cnt = hist[d];
                uint32_t r = 0;
                for (uint32_t q = a; q < a + cnt; ++q) {
                    const uint32_t kq = s_keys[q];
                    r += kq < k ? 1u : 0u;
                    if (kq == k) r += s_vals[q] < me ? 1u : 0u;   // equal keys: by instance number = in input order
                }
                s_fin[a + r] = me;
            }
            __syncthreads();
            epilogue(s_fin, n);
            return;
        }
        __syncthreads();   // (the passes' tables overlay the distribution sort's)
        // a clustered range: stable least-significant-digit passes -- over the instance numbers first (the scatter left the
        // members of a bucket in no particular order), then over the relative keys
        for (int pass = 0; pass < IL.npasses + L.npasses; ++pass) {
            const bool by_inst = pass < IL.npasses;
            const int shift = by_inst ? IL.shift(pass) : L.shift(pass - IL.npasses);
            const uint32_t mask = (1u << (by_inst ? IL.width(pass) : L.width(pass - IL.npasses))) - 1u;
#pragma unroll
            for (int w = 0; w < NW; ++w) s_cnt[w][threadIdx.x] = 0u;
            __syncthreads();   // (also: everybody has read the previous pass' s_keys / s_vals)
#pragma unroll
            for (int i = 0; i < kRangeItems; ++i) dig[i] = ((valid >> i) & 1u) ? ((by_inst ? val[i] : key[i]) >> shift) & mask : 0u;
            wave_rank<kRangeItems, kDepthDigitBits, uint32_t>(dig, valid, s_cnt[wave], rank, lane);
            __syncthreads();
            {
                uint32_t my_tot = 0, run = 0;
                uint32_t c[NW];
#pragma unroll
                for (int w = 0; w < NW; ++w) { c[w] = s_cnt[w][threadIdx.x]; my_tot += c[w]; }
#pragma unroll
                for (int w = 0; w < NW; ++w) { s_cnt[w][threadIdx.x] = run; run += c[w]; }
                uint32_t total;
                const uint32_t incl = block_incl_scan<NW>(my_tot, s_wave, &total);
                s_dstart[threadIdx.x] = incl - my_tot;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < kRangeItems; ++i) {
                if ((valid >> i) & 1u) {
                    const uint32_t pos = s_dstart[dig[i]] + s_cnt[wave][dig[i]] + rank[i];
                    s_keys[pos] = key[i];
                    s_vals[pos] = val[i];
                }
            }
            __syncthreads();
            if (pass + 1 < IL.npasses + L.npasses) {
#pragma unroll
                for (int i = 0; i < kRangeItems; ++i) {
                    const uint32_t loc = wbase + i * 64 + lane;
                    if (loc < wend) { key[i] = s_keys[loc]; val[i] = s_vals[loc]; }
                }
            }
        }
        // (a layout without a digit cannot happen: depth_layout_from gives at least one pass)
        epilogue(s_vals, n);
        return;
    }
    // ---- a range that does not fit: the same passes through memory, one chunk of kMsdCap elements after the other, this
    // workgroup alone (source: the range's stretch of the scattered elements; ping-pong partner: the same stretch of the
    // input buffer, which nobody reads any more).  The release / acquire pair between two passes makes this CU's L1 forget
    // the lines of the stretch it is about to read again.
    if (threadIdx.x == 0) atomicAdd(&counters->reserved[2], 1u);
    const uint2* src = msd_sorted + r0;
    uint2* dst = scratch + r0;
    uint2* const other = const_cast<uint2*>(msd_sorted) + r0;
    for (int pass = 0; pass < IL.npasses + L.npasses; ++pass) {
        const bool by_inst = pass < IL.npasses;
        const int shift = by_inst ? IL.shift(pass) : L.shift(pass - IL.npasses);
        const uint32_t mask = (1u << (by_inst ? IL.width(pass) : L.width(pass - IL.npasses))) - 1u;
        const bool last = pass == IL.npasses + L.npasses - 1;
        auto digit = [&](uint2 e) { return ((by_inst ? e.y : rel_key(e.x)) >> shift) & mask; };
        s_dstart[threadIdx.x] = 0u;
        __syncthreads();
        for (uint32_t k = threadIdx.x; k < n; k += kRangeThreads) atomicAdd(&s_dstart[digit(src[k])], 1u);
        __syncthreads();
        {
            const uint32_t tot = s_dstart[threadIdx.x];
            uint32_t all;
            const uint32_t incl = block_incl_scan<NW>(tot, s_wave, &all);
            s_dstart[threadIdx.x] = incl - tot;   // where the digit's run starts (own word: the scan's barriers are enough)
        }
        __syncthreads();
        for (uint32_t c0 = 0; c0 < n; c0 += (uint32_t)kMsdCap) {
            const uint32_t cn = min((uint32_t)kMsdCap, n - c0);
            uint32_t valid = 0u;
#pragma unroll
            for (int w = 0; w < NW; ++w) s_cnt[w][threadIdx.x] = 0u;
#pragma unroll
            for (int i = 0; i < kRangeItems; ++i) {
                const uint32_t loc = wave * (kRangeItems * 64) + i * 64 + lane;   // (n > 4096 here: wbase is this)
                const bool on = loc < cn;
                const uint2 e = on ? src[c0 + loc] : make_uint2(0u, 0u);
                key[i] = e.x; val[i] = e.y;
                dig[i] = on ? digit(e) : 0u;
                valid |= (on ? 1u : 0u) << i;
            }
            __syncthreads();
            wave_rank<kRangeItems, kDepthDigitBits, uint32_t>(dig, valid, s_cnt[wave], rank, lane);
            __syncthreads();
            {
                uint32_t run = s_dstart[threadIdx.x];
#pragma unroll
                for (int w = 0; w < NW; ++w) { const uint32_t c = s_cnt[w][threadIdx.x]; s_cnt[w][threadIdx.x] = run; run += c; }
                s_dstart[threadIdx.x] = run;      // the next chunk's elements of this digit go behind this chunk's
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < kRangeItems; ++i) {
                if ((valid >> i) & 1u) {
                    const uint32_t pos = s_cnt[wave][dig[i]] + rank[i];
                    if (last) inst_sorted[r0 + pos] = val[i];
                    else dst[pos] = make_uint2(key[i], val[i]);
                }
            }
            __syncthreads();
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        src = dst;
        dst = (dst == other) ? scratch + r0 : other;
    }
    if (G.srect) {   // (the scratch stretch IS this range's stretch of the rectangle buffer: free now)
        const bool ok = counters->overflow < 2u;
        for (uint32_t t = threadIdx.x; t < n; t += kRangeThreads) {
            const uint32_t pos = r0 + t;
            const uint2 rc = ok ? G.binfo[inst_sorted[pos]] : make_uint2(0u, 0u);
            G.srect[pos] = rc;
            const uint32_t cnt = (rc.y & 0xFFFFu) * (rc.y >> 16);
            if (cnt) atomicAdd(&G.bsum[pos >> 8], cnt);
            if (G.bcsum) {
                uint32_t sx0, sy0, sw, sh;
                hier_coarse_rect(rc, sx0, sy0, sw, sh);
                if (sw * sh) atomicAdd(&G.bcsum[pos >> 8], sw * sh);
            }
        }
    }
}

// ---------------------------------------------------------------- split tile sort (a6 + a7)
// Depth keys of the instances as (key, instance) elements: culled instances get the largest key so they sort to the end.
__global__ void __launch_bounds__(256) depth_keys_kernel(int64_t I, const float* depth, const int* radii, uint2* pairs,
                                                         unsigned long long* bits) {
    // 1024 instances per workgroup: the OR of the keys ends in two atomics per workgroup on one of 16 copies -- with one
    // workgroup per 256 instances four thousand of them queued on those words for 17 us (c3)
    __shared__ uint32_t s_two[2];
    if (threadIdx.x < 2) s_two[threadIdx.x] = 0u;
    __syncthreads();
    uint32_t o = 0u, nz = 0u;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = (int64_t)blockIdx.x * 1024 + k * 256 + threadIdx.x;
        if (i < I) {
            const bool visible = radii[i] > 0;
            const uint32_t key = visible ? __float_as_uint(depth[i]) : 0xFFFFFFFFu;
            pairs[i] = make_uint2(key, (uint32_t)i);
            if (visible) { o |= key; nz |= ~key; }
        }
    }
    // which bits vary (tag 0: bin_prepare_kernel zeroed the words): the ONE writer of the depth-bits words, fed with the
    // OR of this thread's four keys and of their complements
    depth_bits_accumulate2(o, nz, bits, 0u, s_two);
}

// Tile rectangles of the instances gathered into depth order (one 8-byte gather per instance) ahead of the emission --
// the path of frames with fewer than 2^21 instances, see emit_pairs_kernel.
// ... and the pair count of every 256 of them (the emission's block size), summed in the same launch (round 4: one launch
// less than gather + scan_reduce256_kernel; -3 us at c3)
__global__ void __launch_bounds__(256) gather_binfo_kernel(int64_t I, const uint32_t* inst_sorted, const uint2* binfo,
                                                           uint2* srect, const hs_counters* counters, uint32_t* block_sums) {
    __shared__ uint32_t s_wave[4];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    uint2 rc = make_uint2(0u, 0u);
    if (i < I) {
        if (counters->overflow < 2u) rc = binfo[inst_sorted[i]];
        srect[i] = rc;
    }
    uint32_t total;
    block_incl_scan((rc.y & 0xFFFFu) * (rc.y >> 16), s_wave, &total);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

// ---- chained scan of the emission (one 64-bit status word per 256-instance block: flag << 62 | pairs) ----
// Two ways to the block-exclusive pair offsets of the emission, chosen by the instance count (measured, binning stage, ms):
//   * ahead of it: gather_binfo_kernel + scan_reduce256_kernel + scan_spine_kernel (three small launches, no chain) --
//     c3 (1 M instances) 0.222, c4 (8 M) 1.716, c2 (0.1 M) 0.114;
//   * inside it: the emission gathers its rectangles itself and learns the earlier blocks' pairs by decoupled look-back
//     -- c3 0.229 (the chain over 3906 blocks costs more than two tiny kernels), c4 1.581 (three passes over 8 M
//     instances and 130 MB of intermediate arrays less), c2 0.111.
// (the switch: scan_in_emission(I), api.hip -- from 2^21 instances on; HS_SCAN_IN_EMISSION=0/1 forces it)
constexpr uint64_t kScFlag = 3ull << 62, kScAgg = 1ull << 62, kScIncl = 2ull << 62, kScPoison = 3ull << 62;
__device__ __forceinline__ void sc_publish(uint64_t* p, uint64_t v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ uint64_t sc_read(const uint64_t* p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// duplicateWithKeys walking the instances in depth order.  A wave owns 64 consecutive instances, whose pair slots
// form ONE contiguous range; its lanes walk that range slot by slot (fully coalesced 8-byte stores), finding each
// slot's owner by binary search over the 64 slot starts in LDS and its tile from the owner's rectangle (row-major, the
// published emission order).  Also records where each instance's slots start (render-backward addresses its
// gradient records with it; the segmented sum walks them).
// The inclusive scan of the depth-ordered pair counts (a5 on the order the pairs are laid out in) is finished HERE.  Frames
// of fewer than 2^21 instances: three small kernels ahead of this one leave the gathered rectangles in `srect`
// and the exclusive sum of every 256-instance block in `block_excl`.  Larger frames (several poses): both are null -- a
// block gathers the tile rectangles of its 256 instances itself (one 8-byte gather each), scans their pair counts, and
// learns the pairs of all earlier blocks by decoupled look-back over one 64-bit status word per block (wave 0 reads 64
// predecessors per step; same progress argument and the same bounded waits as the radix passes).  The LAST block knows num_rendered and publishes
// the verdict for the later kernels: n_sort = R, or 0 when R exceeds the binning capacity (or a wait gave up) -- then
// nothing is sorted, the frame renders empty, and the host sees counters.overflow and replays with a larger capacity
// (blocks that lie below the capacity have emitted their pairs by then: harmless, they are never looked at).
// COUNT: the tile sort of a small frame is a counting sort (tile_colscan_kernel / tile_scatter_kernel below): instead of the
// radix passes' digit totals the workgroup leaves, per (pose, tile) key, how many pairs EACH OF ITS FOUR WAVES emitted --
// four byte fields of one word (a wave's 64 instances touch a tile at most once each) -- as row blockIdx of `tile_counts`.
template <bool BIG, bool COUNT = false>   // BIG: the large-frame path (srect / block_excl null, non-temporal hints: see the rectangle gather below)
__global__ void __launch_bounds__(256) emit_pairs_kernel(int64_t I, int P, int gx, int gy, float4* rec,
                                                         const uint32_t* inst_sorted, const uint2* binfo, const uint2* srect,
                                                         const uint32_t* block_excl, uint64_t* scan_status,
                                                         uint32_t* offs_sorted, uint2* pairs,
                                                         uint8_t* pair_flags, hs_counters* counters, uint64_t capacity,
                                                         uint32_t* ghist, int nbits, int passes,
                                                         unsigned long long* depth_bits, int excl_ready,
                                                         uint32_t* tile_counts = nullptr, int vtiles = 0) {
    // digit totals of the tile sort's passes (<= 4), this block's pairs; COUNT: pairs per key, one byte field per wave
    __shared__ uint32_t s_hist[COUNT ? kCountTilesMax : 4 * 256];
    __shared__ uint32_t s_beg[4][64];
    __shared__ uint4 s_own[4][64];         // per instance: key of its first tile, rectangle width, 1 / width (float), instance
    __shared__ uint32_t s_wsum[4];
    __shared__ unsigned long long s_wsum64[4];
    __shared__ uint64_t s_excl;            // pairs of all earlier blocks; kScPoison in the flag bits: a wait gave up
    for (int t = threadIdx.x; t < (COUNT ? vtiles : passes * 256); t += 256) s_hist[t] = 0;
    // the depth sort is over: its tagged depth-bits words go back to empty, so a replayed graph (same tag every frame)
    // starts each frame's OR from nothing instead of from every earlier frame's bits
    if (blockIdx.x == 0 && threadIdx.x < 2 * kDepthBitsCopies) depth_bits[threadIdx.x] = 0ull;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int blk = blockIdx.x;
    const int64_t i = (int64_t)blk * 256 + threadIdx.x;
    uint32_t inst = 0;
    uint2 rc = make_uint2(0u, 0u);
    // (a depth sort that gave up -- overflow = 2 -- left no instance list: every instance counts as culled)
    if (i < I && counters->overflow < 2u) {
        inst = inst_sorted[i];
        // (srect: the rectangles gathered into depth order by a kernel of their own; else -- frames of millions of instances
        // -- one 8-byte gather each.  That path marks its scattered slot-start store and its streaming pair store
        // non-temporal (HS_EMIT_NT_MASK): at that size neither line is touched again before it would be evicted anyway, and
        // keeping them out of the L2 takes ~120 us off c4's 590 us emission; at c3, where the records stay cached for the
        // render forward, the same hint on the slot start COSTS 10 us)
        if constexpr (BIG && (HS_EMIT_NT_MASK & 1)) rc = make_uint2(__builtin_nontemporal_load(&binfo[inst].x), __builtin_nontemporal_load(&binfo[inst].y));
        else if constexpr (BIG) rc = binfo[inst];
        else rc = srect[i];
    }
    const uint32_t cnt = (rc.y & 0xFFFFu) * (rc.y >> 16);
    uint32_t block_total;
    const uint32_t incl = block_incl_scan(cnt, s_wsum, &block_total);
    if constexpr (!BIG) {
        // the pair counts of all 256-instance blocks were summed ahead of this launch (gather_binfo_kernel); every block adds
        // up the ones in front of it itself -- at most 8192 of them below 2^21 instances, 32 coalesced loads per thread --
        // instead of waiting for a single-workgroup scan kernel in between (round 4: one launch less, -3 us at c3)
        // (beyond 8192 blocks -- a frame of millions of instances forced onto this path -- a scan kernel has turned the sums
        // into exclusive prefixes: `excl_ready`)
        if (excl_ready) {
            if (threadIdx.x == 0) s_excl = block_excl[blk];
        } else {
            // (64-bit: a frame whose pair count wraps 2^32 must still trip the `bend > capacity` check below)
            unsigned long long part = 0;
            for (int j = threadIdx.x; j < blk; j += 256) part += block_excl[j];
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) part += (unsigned long long)__shfl_xor((long long)part, d);
            if ((threadIdx.x & 63) == 0) s_wsum64[threadIdx.x >> 6] = part;
            __syncthreads();
            if (threadIdx.x == 0) s_excl = (s_wsum64[0] + s_wsum64[1]) + (s_wsum64[2] + s_wsum64[3]);
        }
    } else if (wave == 0) {
        if (lane == 0) sc_publish(scan_status + blk, (blk == 0 ? kScIncl : kScAgg) | (uint64_t)block_total);
        uint64_t excl = 0;
        bool poison = false;
        for (int base = blk - 1; base >= 0; base -= 64) {
            const int idx = base - lane;
            uint64_t x = idx >= 0 ? sc_read(scan_status + idx) : kScIncl;
            int polls = 0;
#pragma clang loop unroll(disable)
            while (__ballot((x & kScFlag) == 0ull) != 0ull && polls < kSpinLimitBlockIdx) {   // (wave-uniform: see the radix passes)
                ++polls;
                if (polls == 128) {
                    // helping, as in the radix passes: a silent predecessor may not have started; its pair count depends on
                    // the sorted instance list alone, so this wave adds it up itself (256 rectangles), offers it, walks on
                    uint64_t missing = __ballot((x & kScFlag) == 0ull);
                    while (missing) {
                        const int l = (int)__builtin_ctzll(missing);
                        missing &= missing - 1ull;
                        const int q = base - l;
                        uint32_t tot = 0;
                        for (int r = 0; r < 4; ++r) {
                            const int64_t ii = (int64_t)q * 256 + r * 64 + lane;
                            if (ii < I && counters->overflow < 2u) {
                                const uint2 rq = binfo[inst_sorted[ii]];
                                tot += (rq.y & 0xFFFFu) * (rq.y >> 16);
                            }
                        }
#pragma unroll
                        for (int d = 32; d >= 1; d >>= 1) tot += (uint32_t)__shfl_xor((int)tot, d);
                        if (lane == 0) {
                            atomicCAS(reinterpret_cast<unsigned long long*>(scan_status + q), 0ull, (unsigned long long)(kScAgg | (uint64_t)tot));
                            atomicAdd(&counters->reserved[4], 1u);
                        }
                    }
                }
                __builtin_amdgcn_s_sleep(1);
                if ((x & kScFlag) == 0ull) x = sc_read(scan_status + idx);
            }
            const uint64_t flag = x & kScFlag;
            poison = poison || __ballot(flag == 0ull || flag == kScPoison) != 0ull;
            // the nearest inclusive prefix of this window (lowest lane) ends the walk: everything in front of it counts
            const uint64_t incl_mask = __ballot(flag == kScIncl || flag == kScPoison || flag == 0ull);
            const int first = incl_mask ? (int)__builtin_ctzll(incl_mask) : 64;
            uint64_t v = lane <= first ? (x & ~kScFlag) : 0ull;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1) v += (uint64_t)__shfl_xor((long long)v, d);
            excl += v;
            if (incl_mask) break;
        }
        if (lane == 0) {
            if (blk != 0) sc_publish(scan_status + blk, (poison ? kScPoison : kScIncl) | ((excl + block_total) & ~kScFlag));
            s_excl = poison ? kScPoison : excl;
        }
    }
    __syncthreads();
    const bool poison = (s_excl & kScFlag) != 0ull;
    const uint64_t bexcl = s_excl & ~kScFlag;
    const uint64_t bend = bexcl + block_total;
    if (blk == (int)gridDim.x - 1 && threadIdx.x == 0) {   // the last block knows R: the verdict for the later kernels
        const bool over = bend > capacity;
        counters->num_rendered = bend > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)bend;
        if (poison) counters->overflow = 2u;
        else if (over && counters->overflow < 2u) counters->overflow = 1u;
        counters->reserved[0] = (poison || over || counters->overflow != 0u) ? 0u : (uint32_t)bend;
        counters->reserved[5] = COUNT ? (uint32_t)kTileSortCount : (uint32_t)kTileSortRadix;   // which tile sort follows
    }
    uint32_t beg = 0, end = 0;
    end = (uint32_t)bexcl + incl;   // (32 bits: exact whenever the block lies below the capacity < 2^30)
    beg = end - cnt;
    if (i < I) offs_sorted[i] = end;   // written even on overflow: the segmented sum then finds nothing flagged
    if (poison || bend > capacity) return;
    if (i < I && end > beg) {
        float* slot_start = &reinterpret_cast<float*>(rec + kRecF4 * (int64_t)inst + 2)[3];
        if constexpr (BIG && (HS_EMIT_NT_MASK & 2)) __builtin_nontemporal_store(__uint_as_float(beg), slot_start);
        else *slot_start = __uint_as_float(beg);
    }
    // lanes past the end of the array inherit the running end so the range stays monotone
    const uint32_t last_end = __shfl(end, 63 - __builtin_clzll(__ballot(i < I) | 1ull));
    if (i >= I) beg = end = last_end;
    // what a slot needs from its owner, once per instance instead of once per pair: the key of the rectangle's first tile
    // (pose * tiles + row * gx + column), the rectangle's width and its reciprocal.  slot -> (row, column) inside the
    // rectangle: row = floor(t / w) as (uint)((t + 0.5) * fl(1 / w)) -- exact for t < 2^22 (the product is off by at most
    // (t + 0.5) / w * 2^-23, the quotient's fraction stays 0.5 / w away from an integer; hs_plan keeps a frame below 2^22
    // tiles), four instructions instead of the thirty of an integer division.
    {
        const uint32_t w = rc.y & 0xFFFFu;
        const uint32_t pose = I > (int64_t)P ? inst / (uint32_t)P : 0u;
        const uint32_t key0 = pose * (uint32_t)(gx * gy) + (rc.x >> 16) * (uint32_t)gx + (rc.x & 0xFFFFu);
        s_beg[wave][lane] = beg;
        s_own[wave][lane] = make_uint4(key0, w, __float_as_uint(1.0f / (float)w), inst);
    }
    // digit layout of the tile sort's passes (same as radix_sort_packed), once
    int sh[4] = {0, 0, 0, 0}, wd[4] = {0, 0, 0, 0};
    {
        int shift = 0;
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            if (pass < passes) {
                const int w = (nbits - shift + (passes - pass) - 1) / (passes - pass);
                sh[pass] = shift; wd[pass] = w;
                shift += w;
            }
        }
    }
    const uint32_t first = __shfl(beg, 0);
    const uint32_t total = last_end - first;
    // same wave wrote and reads these LDS rows: no workgroup barrier needed (wave-private rows)
    for (uint32_t p = lane; p < total; p += 64) {
        const uint32_t pos = first + p;
        uint32_t k = 0;
#pragma unroll
        for (uint32_t step = 32; step >= 1; step >>= 1)   // largest k with beg[k] <= pos (k + step <= 63); slots < 2^30
            k |= (uint32_t)((int32_t)(s_beg[wave][k + step] - pos - 1u) >> 31) & step;
        const uint4 o = s_own[wave][k];
        const uint32_t t = pos - s_beg[wave][k];
        const uint32_t ty = (uint32_t)(((float)t + 0.5f) * __uint_as_float(o.z));
        const uint32_t tx = t - ty * o.y;
        const uint32_t key = o.x + ty * (uint32_t)gx + tx;
        if constexpr (BIG && (HS_EMIT_NT_MASK & 4)) { __builtin_nontemporal_store(key, &pairs[pos].x); __builtin_nontemporal_store(o.w, &pairs[pos].y); }
        else pairs[pos] = make_uint2(key, o.w);
        pair_flags[pos] = 0;  // "gradient record written" flag of this slot, set by the render backward
        if constexpr (COUNT) {
            atomicAdd(&s_hist[key], 1u << (8 * wave));
            continue;
        }
        // digit totals for the single-sweep tile sort.  The lanes of a wave hold neighbouring tiles of a few Gaussians:
        // when they all agree on a digit, one lane adds the count
        const uint64_t act = __ballot(true);
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
            if (pass < passes) {
                const uint32_t dgt = (key >> sh[pass]) & ((1u << wd[pass]) - 1u);
                const uint32_t d0 = __builtin_amdgcn_readfirstlane(dgt);
                if (__ballot(dgt != d0) == 0ull) {
                    if (lane == (int)__builtin_ctzll(act)) atomicAdd(&s_hist[pass * 256 + d0], (uint32_t)__popcll(act));
                } else {
                    atomicAdd(&s_hist[pass * 256 + dgt], 1u);
                }
            }
        }
    }
    __syncthreads();
    if constexpr (COUNT) {
        uint32_t* row = tile_counts + (int64_t)blockIdx.x * vtiles;
        for (int t = threadIdx.x; t < vtiles; t += 256) row[t] = s_hist[t];
        return;
    }
    uint32_t* mine = ghist + (blockIdx.x % kGhistCopies) * (8 * 256);
    for (int t = threadIdx.x; t < passes * 256; t += 256) {
        const uint32_t c = s_hist[t];
        if (c) atomicAdd(&mine[t], c);
    }
}

// ---------------------------------------------------------------- tile sort of small frames by counting (a7 + a8)
// A frame of few tiles and few instances (count_sort_fits, hs_common.h: <= 4096 (pose, tile) keys, <= 2^21 entries in the
// matrix below; BASELINE c2 = 2500 tiles x 391 emission workgroups) does not need radix passes over its pairs: the pair
// emission walks the instances in depth order, so the sorted position of a pair is
//     start of its tile + pairs of that tile emitted by EARLIER workgroups + ... by earlier waves of its workgroup
//     + ... earlier in its own wave's walk,
// i.e. a counting sort by the whole tile id, stable by construction.  Three kernels instead of emission + two radix passes
// + tile ranges (each bound by its chain of dependent round trips at this size, not by bytes):
//   1. emit_pairs_kernel<., COUNT>: row b of `counts` = pairs per tile of emission workgroup b, one byte per wave;
//   2. tile_colscan_kernel: per tile, the exclusive prefix of the rows' totals down the column (`bases`) and the column
//      total (`totals`);
//   3. tile_scatter_kernel: workgroup b turns (scan of `totals`) + bases[b] + its waves' byte fields into one running
//      position per (wave, tile) in LDS, and its waves re-walk their pairs (still in the emission's buffer, depth order)
//      in the emission's order, handing out positions: point_list; workgroup 0 also writes the tile ranges
//      (start, start + total) and the host copy of the frame's counters.  (The sorted tile ids -- keys_sorted, which the
//      radix path needs for its tile ranges -- are nobody's input here: a second scattered 4-byte store per pair took the
//      kernel from 15 to 22 us, so the inspection stage writes them from the ranges when somebody asks: tile_keys_kernel.)
// No look-back chain, no status words, nothing to clear; results are bit for bit the radix path's (tests run both).
constexpr int kColscanCols = 16, kColscanSlots = 64;   // a workgroup of tile_colscan_kernel: 16 columns x 64 row slots

__device__ __forceinline__ uint32_t bytes_sum(uint32_t v) { return (v & 0xFFu) + ((v >> 8) & 0xFFu) + ((v >> 16) & 0xFFu) + (v >> 24); }

__global__ void __launch_bounds__(kColscanCols * kColscanSlots) tile_colscan_kernel(const uint32_t* counts, int nrows, int vtiles,
                                                                                const uint32_t* n_sort, uint32_t* bases,
                                                                                uint32_t* totals) {
    __shared__ uint32_t s_wsum[kColscanSlots / 4][kColscanCols];   // per wave (four row slots) and column
    if (*n_sort == 0u) return;   // nothing to sort (no pairs, or the frame overflowed its capacity: rows may be missing)
    const int c = threadIdx.x % kColscanCols, rs = threadIdx.x / kColscanCols;
    const int t = blockIdx.x * kColscanCols + c;
    const int rps = (nrows + kColscanSlots - 1) / kColscanSlots;          // consecutive rows per slot
    const int r0 = min(nrows, rs * rps), r1 = min(nrows, r0 + rps);
    const bool on = t < vtiles;
    // (up to eight rows per slot -- 512 emission workgroups, 131 k instances -- stay in registers for the second walk)
    constexpr int KEEP = 8;
    const bool keep = rps <= KEEP;
    uint32_t kept[KEEP];
    uint32_t sum = 0;
    if (on && keep) {
#pragma unroll
        for (int j = 0; j < KEEP; ++j) {
            kept[j] = r0 + j < r1 ? counts[(int64_t)(r0 + j) * vtiles + t] : 0u;
            sum += bytes_sum(kept[j]);
        }
    } else if (on) {
#pragma unroll 8
        for (int r = r0; r < r1; ++r) sum += bytes_sum(counts[(int64_t)r * vtiles + t]);
    }
    // rows above this thread's, same column: the wave's lanes are (row slot % 4, column) = (lane / 16, lane % 16) -- two
    // shuffle steps inside the wave, then the sums of the waves in front
    static_assert(kColscanCols == 16, "lane = 16 * (row slot % 4) + column");
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t incl = sum;
    { const uint32_t u = __shfl_up(incl, 16); if (lane >= 16) incl += u; }
    { const uint32_t u = __shfl_up(incl, 32); if (lane >= 32) incl += u; }
    if (lane >= 48) s_wsum[wave][c] = incl;
    __syncthreads();
    uint32_t run = incl - sum, total = 0;
#pragma unroll
    for (int w = 0; w < kColscanSlots / 4; ++w) {
        const uint32_t v = s_wsum[w][c];
        if (w < wave) run += v;
        total += v;
    }
    if (!on) return;
    if (rs == 0) totals[t] = total;
    if (keep) {
#pragma unroll
        for (int j = 0; j < KEEP; ++j) {
            if (r0 + j < r1) bases[(int64_t)(r0 + j) * vtiles + t] = run;
            run += bytes_sum(kept[j]);
        }
        return;
    }
#pragma unroll 8
    for (int r = r0; r < r1; ++r) {   // (second read of the rows: L2 hits)
        const uint32_t v = counts[(int64_t)r * vtiles + t];
        bases[(int64_t)r * vtiles + t] = run;
        run += bytes_sum(v);
    }
}

__global__ void __launch_bounds__(256) tile_scatter_kernel(int64_t I, int vtiles, const uint2* pairs, const uint32_t* offs_sorted,
                                                           const uint32_t* counts, const uint32_t* bases, const uint32_t* totals,
                                                           const uint32_t* n_sort, uint32_t* point_list,
                                                           uint2* ranges, const hs_counters* counters, uint32_t* counters_host) {
    extern __shared__ uint32_t s_pos[];      // [4][vtiles]: next position of (wave, key); [vtiles] more: tile starts
    __shared__ uint32_t s_wave[4];
    if (counters_host && blockIdx.x == 0 && threadIdx.x < 8)   // (what tile_ranges_kernel does on the radix path)
        __hip_atomic_store(counters_host + threadIdx.x, reinterpret_cast<const uint32_t*>(counters)[threadIdx.x],
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (*n_sort == 0u) return;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int blk = blockIdx.x;
    // Tile starts = exclusive scan of the column totals.  Everything is fetched with coalesced loads issued together (key
    // k = thread + 256 j: the totals, this workgroup's rows of `counts` and `bases` -- up to 48 loads in flight per
    // thread); the totals pass through LDS so that a thread can scan E CONSECUTIVE keys.
    constexpr int JMAX = kCountTilesMax / 256;
    uint32_t* const s_start = s_pos + 4 * vtiles;
    const uint32_t* crow = counts + (int64_t)blk * vtiles;
    const uint32_t* brow = bases + (int64_t)blk * vtiles;
    uint32_t tv[JMAX], cv[JMAX], bv[JMAX];
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const int k = (int)threadIdx.x + 256 * j;
        const bool in = k < vtiles;
        tv[j] = in ? totals[k] : 0u; cv[j] = in ? crow[k] : 0u; bv[j] = in ? brow[k] : 0u;
    }
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const int k = (int)threadIdx.x + 256 * j;
        if (k < vtiles) s_start[k] = tv[j];
    }
    __syncthreads();
    {
        const int E = (vtiles + 255) / 256;   // <= 16 consecutive keys per thread
        const int k0 = min(vtiles, (int)threadIdx.x * E), k1 = min(vtiles, k0 + E);
        uint32_t mine = 0;
        for (int k = k0; k < k1; ++k) mine += s_start[k];
        uint32_t all;
        uint32_t start = block_incl_scan(mine, s_wave, &all) - mine;   // (its barriers order the reads above before the writes below)
        for (int k = k0; k < k1; ++k) { const uint32_t tot = s_start[k]; s_start[k] = start; start += tot; }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < JMAX; ++j) {
        const int k = (int)threadIdx.x + 256 * j;
        if (k < vtiles) {
            const uint32_t start = s_start[k], v = cv[j];
            uint32_t p = start + bv[j];
            s_pos[k] = p;                 p += v & 0xFFu;
            s_pos[vtiles + k] = p;        p += (v >> 8) & 0xFFu;
            s_pos[2 * vtiles + k] = p;    p += (v >> 16) & 0xFFu;
            s_pos[3 * vtiles + k] = p;
            if (blk == 0 && tv[j]) ranges[k] = make_uint2(start, start + tv[j]);   // (tiles without pairs keep their cleared (0, 0))
        }
    }
    __syncthreads();
    // this wave's slots: those of its 64 instances, [end of the instance before them, end of their last)
    const int64_t i_first = (int64_t)blk * 256 + wave * 64;
    if (i_first >= I) return;
    const int64_t i_last = min(I, i_first + 64) - 1;
    const uint32_t ws = i_first ? offs_sorted[i_first - 1] : 0u, we = offs_sorted[i_last];
    uint32_t* const pos_w = s_pos + wave * vtiles;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    // (four rounds of 64 pairs requested at once: the walk is a chain of dependent LDS updates, its loads need not be)
    for (uint32_t base4 = ws; base4 < we; base4 += 256) {
        uint2 e4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t p = base4 + 64 * q + lane;
            e4[q] = p < we ? pairs[p] : make_uint2(0u, 0u);
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const uint32_t base = base4 + 64 * q;
            if (base >= we) break;
            const uint2 e = e4[q];
            const bool valid = base + lane < we;
            uint64_t peers = __ballot(valid);   // match-any: lanes holding the same key
#pragma unroll
            for (int b = 0; b < 12; ++b) {
                const uint64_t m = __ballot((e.x >> b) & 1u);
                peers &= ((e.x >> b) & 1u) ? m : ~m;
            }
            if (valid) {
                const uint32_t before = pos_w[e.x];                       // (all peers read before the last one writes: same wave)
                const uint32_t dst = before + (uint32_t)__popcll(peers & lt_mask);
                if ((peers >> lane) == 1ull) pos_w[e.x] = dst + 1u;
                point_list[dst] = e.y;
            }
        }
    }
}

// keys_sorted of a frame sorted by counting, from its tile ranges (HS_STAGE_OFFSETS, inspection only): one wave per tile
__global__ void __launch_bounds__(256) tile_keys_kernel(const uint2* ranges, int vtiles, uint32_t capacity, uint32_t* keys_sorted,
                                                        const hs_counters* counters) {
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    // (the radix passes wrote the sorted tile ids themselves; which sort the frame had is on the device, so an inspection
    // call need not ask the environment again -- it may have changed since the forward)
    if (t >= vtiles || counters->reserved[5] == (uint32_t)kTileSortRadix) return;
    const uint2 r = ranges[t];
    for (uint32_t p = r.x + (threadIdx.x & 63); p < min(r.y, capacity); p += 64) keys_sorted[p] = (uint32_t)t;
}

// ---------------------------------------------------------------- hierarchical tile sort (a6 + a7 + a8)
// Coarse stable pass + on-chip expansion (round 6).  The radix tile sort moves every (tile, instance)
// pair -- 8 bytes -- through two passes (read + write each) after writing it once: 5 x 8 x R bytes, R = 6.8 M at c3.  A
// Gaussian's tile rectangle is a few tiles wide, so the pairs of one instance that fall into the same 8 x 8-tile SUPER-TILE
// are described by ONE element: (super-tile id, rectangle clipped to the super-tile, instance) -- 1.5 M elements at c3.
//   1. hier_gather_kernel: the rectangles gathered into depth order and, per 256 instances, the sums of their pair counts
//      and of their element counts (the scan of both is finished by the emission, as on the radix path);
//   2. hier_emit_kernel: walks the instances in depth order and writes their elements (a wave's 64 instances own one
//      contiguous range of element slots; lanes walk it slot by slot); leaves the inclusive pair offsets in depth order
//      and every instance's first pair slot (the render backward addresses its gradient records with it -- the slot of a
//      pair is slot start + position of the tile inside the rectangle, nothing the sort decides), clears the pair flags,
//      counts the elements per super-tile and the digit totals of the pass(es) below, and verdicts num_rendered;
//   3. radix_sweep_kernel over the elements, by super-tile id only: ONE pass for up to 256 (pose, super-tile) keys (c3:
//      135), two up to 65 536; stable, so every super-tile's elements stay in depth order;
//   4. hier_plan_kernel (one workgroup): first element of every super-tile, its chunks of hier_chunk(I) = 512 / 1024 elements, one
//      descriptor per chunk;
//   5. hier_count_kernel, one workgroup per chunk: pairs per tile of the super-tile and per wave (a wave takes 256
//      consecutive elements: four +-1 corner marks per element into a 9 x 9 grid, then a 2-D prefix sum), and the tile
//      totals (atomics: a dozen per address);
//   6. hier_tiles_kernel (one workgroup per 1024 tiles): exclusive scan of the tile totals in tile-id order = ranges, first positions;
//   7. hier_scatter_kernel, one workgroup per chunk: first position of (wave, tile) = start of the tile + pairs of the
//      tile in the super-tile's earlier chunks (summed here from their rows) + in earlier waves; the waves then walk the
//      pairs of their elements in element order (row-major inside the clipped rectangle) and hand out positions by
//      match-any over the six bits of the tile inside the super-tile: point_list.
// Stable by construction: depth order -> stable pass -> chunks, waves and the walk in element order.  keys_sorted has no
// reader (HS_STAGE_OFFSETS fills it from the ranges on request, as for the counting sort).
#ifndef HS_TUNE_HIER_GRID
#define HS_TUNE_HIER_GRID 256
#endif
constexpr int kHierGridPerXcd = HS_TUNE_HIER_GRID;   // expansion workgroups per XCD (persistent: each walks its XCD's chunks)
#ifndef HS_TUNE_HIER_STAGE
#define HS_TUNE_HIER_STAGE 1
#endif
constexpr int kHierStage = 1024;   // pairs of one round (64 elements) a wave stages in LDS (8 KB per wave)
// ablation switches of hier_emit_kernel (timing experiments only; the result is wrong with any of them set)
#ifndef HS_ABL
#define HS_ABL 0
#endif

// rectangle (x0 | y0 << 16, w | h << 16) -> its super-tile rectangle: first super-tile column / row and the numbers of them
__device__ __forceinline__ void hier_coarse_rect(uint2 rc, uint32_t& sx0, uint32_t& sy0, uint32_t& sw, uint32_t& sh) {
    const uint32_t x0 = rc.x & 0xFFFFu, y0 = rc.x >> 16, w = rc.y & 0xFFFFu, h = rc.y >> 16;
    sx0 = x0 / kSuper; sy0 = y0 / kSuper;
    const bool any = w != 0u && h != 0u;
    sw = any ? (x0 + w - 1u) / kSuper - sx0 + 1u : 0u;
    sh = any ? (y0 + h - 1u) / kSuper - sy0 + 1u : 0u;
}

// NT = 256: one workgroup per emission workgroup, its two sums (c3: 14.7 us; 1024-thread workgroups: 16.5).  NT = 1024
// (frames of >= 2^21 instances): the sums of every 256 AND of the whole 1024, so that an emission workgroup adds up a
// quarter as many words for what lies in front of it (c4: 7812 + 3 instead of 31 250 -- which used to take two single-workgroup
// scan kernels of 32 us each ahead of the emission).
template <int NT>
__global__ void __launch_bounds__(NT) hier_gather_kernel(int64_t I, const uint32_t* inst_sorted, const uint2* binfo,
                                                         uint2* srect, const hs_counters* counters, uint32_t* block_sums,
                                                         uint32_t* block_csums, uint32_t* super_sums, uint32_t* super_csums,
                                                         uint32_t* zero, int64_t n_zero) {
    constexpr int NW = NT / 64, NSUB = NT / 256;
    __shared__ uint32_t s_wave[NW];
    for (int64_t t = (int64_t)blockIdx.x * NT + threadIdx.x; t < n_zero; t += (int64_t)gridDim.x * NT) zero[t] = 0u;
    const int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x;
    uint2 rc = make_uint2(0u, 0u);
    if (i < I) {
        if (counters->overflow < 2u) rc = binfo[inst_sorted[i]];
        srect[i] = rc;
    }
    uint32_t sx0, sy0, sw, sh;
    hier_coarse_rect(rc, sx0, sy0, sw, sh);
    const int64_t nsub = (I + 255) / 256;
    const bool sub_on = (int)threadIdx.x < NSUB && (int64_t)blockIdx.x * NSUB + threadIdx.x < nsub;
    const int w0 = 4 * (int)(threadIdx.x % NSUB);
    uint32_t total;
    block_incl_scan<NW>((rc.y & 0xFFFFu) * (rc.y >> 16), s_wave, &total);     // (leaves the wave totals in s_wave)
    if (sub_on) block_sums[blockIdx.x * NSUB + threadIdx.x] = (s_wave[w0] + s_wave[w0 + 1]) + (s_wave[w0 + 2] + s_wave[w0 + 3]);
    if (NSUB > 1 && threadIdx.x == 0) super_sums[blockIdx.x] = total;
    __syncthreads();
    block_incl_scan<NW>(sw * sh, s_wave, &total);
    if (sub_on) block_csums[blockIdx.x * NSUB + threadIdx.x] = (s_wave[w0] + s_wave[w0 + 1]) + (s_wave[w0 + 2] + s_wave[w0 + 3]);
    if (NSUB > 1 && threadIdx.x == 0) super_csums[blockIdx.x] = total;
}

// Element: .x = (pose, super-tile) key | clipped rectangle above bit `kb` (x0: 3 bits, y0: 3, w - 1: 3, h - 1: 3, all in
// tiles relative to the super-tile), .y = instance.
// Non-temporal hints of the hierarchical emission at >= 2^21 instances (bit 2: slot start, bit 4: elements), measured at c4
// (hier_emit_kernel, us): none 338, both 411 -- unlike the pair emission (HS_EMIT_NT_MASK) the element stream is short and
// is read again at once by the radix pass, so keeping it out of the L2 only hurts.  None it is.
#ifndef HS_HIER_NT_MASK
#define HS_HIER_NT_MASK 0
#endif
template <bool BIG>   // BIG (>= 2^21 instances): the instantiation the hints above apply to
__global__ void __launch_bounds__(256) hier_emit_kernel(int64_t I, int P, int gx, int gy, float4* rec,
                                                        const uint32_t* inst_sorted, const uint2* srect,
                                                        const uint32_t* block_excl, const uint32_t* block_cexcl,
                                                        const uint32_t* super_sums, const uint32_t* super_csums,
                                                        uint32_t* offs_sorted, uint2* elems, uint8_t* pair_flags,
                                                        hs_counters* counters, uint64_t capacity, uint32_t* ghist, int kb,
                                                        int passes, unsigned long long* depth_bits, int excl_ready,
                                                        uint32_t* hier /* header */, uint32_t* st_count, int nst, int nst_pad) {
    __shared__ uint32_t s_hist[kHierStMax];    // elements per (pose, super-tile) key, this workgroup
    __shared__ uint32_t s_dig[4 * 256];        // ... per digit of the pass(es)
    __shared__ uint32_t s_beg[4][64];
    __shared__ uint4 s_own[4][64];             // per instance: key of its first super-tile | columns of super-tiles, rectangle, 1 / columns, instance
    __shared__ uint32_t s_wsum[4];
    __shared__ unsigned long long s_wsum64[4];
    __shared__ uint64_t s_excl;
    __shared__ uint32_t s_cexcl;
    for (int t = threadIdx.x; t < nst; t += 256) s_hist[t] = 0;
    for (int t = threadIdx.x; t < passes * 256; t += 256) s_dig[t] = 0;
    if (blockIdx.x == 0 && threadIdx.x < 2 * kDepthBitsCopies) depth_bits[threadIdx.x] = 0ull;   // (see emit_pairs_kernel)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int blk = blockIdx.x;
    const int64_t i = (int64_t)blk * 256 + threadIdx.x;
    uint32_t inst = 0;
    uint2 rc = make_uint2(0u, 0u);
    if (i < I && counters->overflow < 2u) { inst = inst_sorted[i]; rc = srect[i]; }
    const uint32_t cnt = (rc.y & 0xFFFFu) * (rc.y >> 16);
    uint32_t sx0, sy0, sw, sh;
    hier_coarse_rect(rc, sx0, sy0, sw, sh);
    const uint32_t ccnt = sw * sh;
    uint32_t block_total, block_ctotal;
    const uint32_t incl = block_incl_scan(cnt, s_wsum, &block_total);
    const uint32_t cincl = block_incl_scan(ccnt, s_wsum, &block_ctotal);
    if (excl_ready) {
        if (threadIdx.x == 0) { s_excl = block_excl[blk]; s_cexcl = block_cexcl[blk]; }
    } else {
        unsigned long long part = 0, cpart = 0;    // (64-bit: see emit_pairs_kernel)
        if (super_sums) {
            // (the sums of the 1024-instance groups in front of this workgroup's, then of the 256-instance ones inside its group)
            const int nsup = (HS_ABL & 8) ? 0 : blk / 4;
            for (int j = threadIdx.x; j < nsup; j += 256) { part += super_sums[j]; cpart += super_csums[j]; }
            if ((int)threadIdx.x < blk - 4 * (blk / 4)) { part += block_excl[4 * (blk / 4) + threadIdx.x]; cpart += block_cexcl[4 * (blk / 4) + threadIdx.x]; }
        } else {
            for (int j = threadIdx.x; j < ((HS_ABL & 8) ? 0 : blk); j += 256) { part += block_excl[j]; cpart += block_cexcl[j]; }
        }
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            part += (unsigned long long)__shfl_xor((long long)part, d);
            cpart += (unsigned long long)__shfl_xor((long long)cpart, d);
        }
        if (lane == 0) { s_wsum64[wave] = part; s_beg[0][wave] = (uint32_t)cpart; }
        __syncthreads();
        if (threadIdx.x == 0) {
            s_excl = (s_wsum64[0] + s_wsum64[1]) + (s_wsum64[2] + s_wsum64[3]);
            s_cexcl = (s_beg[0][0] + s_beg[0][1]) + (s_beg[0][2] + s_beg[0][3]);
        }
    }
    __syncthreads();
    const uint64_t bexcl = s_excl;
    const uint64_t bend = bexcl + block_total;
    const uint32_t cbexcl = s_cexcl;
    if (blk == (int)gridDim.x - 1 && threadIdx.x == 0) {   // the last block knows R: the verdict for the later kernels
        const bool over = bend > capacity;
        counters->num_rendered = bend > 0xFFFFFFFFull ? 0xFFFFFFFFu : (uint32_t)bend;
        if (over && counters->overflow < 2u) counters->overflow = 1u;
        const bool empty = over || counters->overflow != 0u;
        counters->reserved[0] = empty ? 0u : (uint32_t)bend;
        counters->reserved[5] = (uint32_t)kTileSortHier;
        hier[0] = empty ? 0u : cbexcl + block_ctotal;      // elements to sort
    }
    const uint32_t end = (uint32_t)bexcl + incl, beg = end - cnt;
    if (i < I) offs_sorted[i] = end;   // written even on overflow: the segmented sum then finds nothing flagged
    if (bend > capacity) return;
    if (!(HS_ABL & 1) && i < I && end > beg) {
        float* slot_start = &reinterpret_cast<float*>(rec + kRecF4 * (int64_t)inst + 2)[3];
        if constexpr (BIG && (HS_HIER_NT_MASK & 2)) __builtin_nontemporal_store(__uint_as_float(beg), slot_start);
        else *slot_start = __uint_as_float(beg);
    }
    // "gradient record written" flags of this workgroup's pair slots [bexcl, bend): bytes up to the first 16-byte boundary,
    // 16-byte stores, bytes behind the last one
    if (!(HS_ABL & 2)) {
        const uint64_t a0 = min(bend, (bexcl + 15ull) & ~15ull), a1 = max(a0, bend & ~15ull);
        for (uint64_t q = bexcl + threadIdx.x; q < a0; q += 256) pair_flags[q] = 0;
        for (uint64_t q = a0 + 16ull * threadIdx.x; q < a1; q += 16ull * 256) *reinterpret_cast<uint4*>(pair_flags + q) = make_uint4(0u, 0u, 0u, 0u);
        for (uint64_t q = a1 + threadIdx.x; q < bend; q += 256) pair_flags[q] = 0;
    }
    // element slots of this wave's 64 instances: one contiguous range, walked slot by slot
    uint32_t cend = cbexcl + cincl, cbeg = cend - ccnt;
    const uint32_t last_end = __shfl(cend, 63 - __builtin_clzll(__ballot(i < I) | 1ull));
    if (i >= I) cbeg = cend = last_end;
    {
        const int sgx = (gx + kSuper - 1) / kSuper, sgy = (gy + kSuper - 1) / kSuper;
        const uint32_t pose = I > (int64_t)P ? inst / (uint32_t)P : 0u;
        const uint32_t key0 = pose * (uint32_t)(sgx * sgy) + sy0 * (uint32_t)sgx + sx0;
        s_beg[wave][lane] = cbeg;
        // (key0 < 2^11, sw <= 2^19 / 8: a frame has fewer than 2^22 tiles per pose)
        s_own[wave][lane] = make_uint4(key0 | (sw << 12), rc.x, rc.y, inst);
    }
    const int sgx = (gx + kSuper - 1) / kSuper;
    const uint32_t first = __shfl(cbeg, 0);
    const uint32_t total = last_end - first;
    for (uint32_t p = lane; p < ((HS_ABL & 4) ? 0u : total); p += 64) {
        const uint32_t pos = first + p;
        uint32_t k = 0;
#pragma unroll
        for (uint32_t step = 32; step >= 1; step >>= 1)   // largest k with beg[k] <= pos
            k |= (uint32_t)((int32_t)(s_beg[wave][k + step] - pos - 1u) >> 31) & step;
        const uint4 o = s_own[wave][k];
        const uint32_t t = pos - s_beg[wave][k];
        const uint32_t csw = o.x >> 12;
        const uint32_t cy = (uint32_t)(((float)t + 0.5f) * (1.0f / (float)csw));   // exact: t < 2^22 (emit_pairs_kernel)
        const uint32_t cx = t - cy * csw;
        const uint32_t key = (o.x & 0xFFFu) + cy * (uint32_t)sgx + cx;
        // the rectangle clipped to this super-tile, in tiles relative to it
        const uint32_t x0 = o.y & 0xFFFFu, y0 = o.y >> 16, x1 = x0 + (o.z & 0xFFFFu), y1 = y0 + (o.z >> 16);
        const uint32_t ox = (x0 / kSuper + cx) * kSuper, oy = (y0 / kSuper + cy) * kSuper;
        const uint32_t lx0 = max(x0, ox) - ox, lx1 = min(x1, ox + kSuper) - ox;
        const uint32_t ly0 = max(y0, oy) - oy, ly1 = min(y1, oy + kSuper) - oy;
        const uint32_t word = key | ((lx0 | (ly0 << 3) | ((lx1 - lx0 - 1u) << 6) | ((ly1 - ly0 - 1u) << 9)) << kb);
        if constexpr (BIG && (HS_HIER_NT_MASK & 4)) { __builtin_nontemporal_store(word, &elems[pos].x); __builtin_nontemporal_store(o.w, &elems[pos].y); }
        else elems[pos] = make_uint2(word, o.w);
        atomicAdd(&s_hist[key], 1u);
    }
    __syncthreads();
    uint32_t* mine = st_count + (int64_t)(blockIdx.x % kHierCopies) * nst_pad;
    for (int t = threadIdx.x; t < nst; t += 256) {
        const uint32_t c = s_hist[t];
        if (c) {
            atomicAdd(&mine[t], c);
            int shift = 0;
            for (int pass = 0; pass < passes; ++pass) {   // digit layout of radix_sort_packed
                const int w = (kb - shift + (passes - pass) - 1) / (passes - pass);
                atomicAdd(&s_dig[pass * 256 + (((uint32_t)t >> shift) & ((1u << w) - 1u))], c);
                shift += w;
            }
        }
    }
    __syncthreads();
    uint32_t* gh = ghist + (blockIdx.x % kGhistCopies) * (8 * 256);
    for (int t = threadIdx.x; t < passes * 256; t += 256) {
        const uint32_t c = s_dig[t];
        if (c) atomicAdd(&gh[t], c);
    }
}

// One workgroup: the sorted elements' layout.  coarse_first[s] = first element of (pose, super-tile) key s (nst + 1
// entries).  The chunks are numbered XCD-class-major: class x = s % 8 first to last, inside a class by s, inside a key by
// position -- so the chunks of one key are consecutive and the expansion workgroups of XCD x (blockIdx % 8 == x) take the
// chunks [xfirst[x], xfirst[x + 1]): every tile list of a super-tile is then written through ONE XCD's L2 (a 128-byte line
// of point_list written by workgroups on several XCDs goes to memory in pieces: 49 us instead of ~20 at c3).
// chunk_first[s] = first chunk of key s, desc[c] = (key, first element, one past the last, chunk index inside the key),
// hier[1] = chunks, hier[8 + x] = xfirst[x] (9 entries).  An empty or overflowed frame has none.
__global__ void __launch_bounds__(1024) hier_plan_kernel(const hs_counters* counters, uint32_t* hier, const uint32_t* st_count,
                                                         int nst, int nst_pad, uint32_t* coarse_first, uint32_t* chunk_first,
                                                         uint4* desc, uint32_t chunk) {
    __shared__ uint32_t s_wave[16];
    __shared__ uint32_t s_cnt[kHierStMax], s_first[kHierStMax];
    // (hier[0]: zeroed by the emission's verdict on overflow and by a radix pass that gave up -- its kill word)
    const bool empty = hier[0] == 0u;
    uint32_t ccarry = 0;
    for (int base = 0; base < nst; base += 1024) {      // natural order: element counts and their exclusive scan
        const int sidx = base + threadIdx.x;
        uint32_t cnt = 0;
        if (sidx < nst && !empty) {
#pragma unroll
            for (int c = 0; c < kHierCopies; ++c) cnt += st_count[(int64_t)c * nst_pad + sidx];
        }
        uint32_t ctot;
        const uint32_t cin = block_incl_scan<16>(cnt, s_wave, &ctot);
        if (sidx < nst) { s_cnt[sidx] = cnt; s_first[sidx] = ccarry + cin - cnt; coarse_first[sidx] = ccarry + cin - cnt; }
        ccarry += ctot;
    }
    __syncthreads();
    const int per = (nst + 7) / 8;                      // keys per class (the last ones of a class may not exist)
    uint32_t kcarry = 0;
    for (int base = 0; base < 8 * per; base += 1024) {  // class-major order: chunk numbers
        const int j = base + threadIdx.x;
        const int x = j / per, sidx = (j - x * per) * 8 + x;
        const bool on = j < 8 * per && sidx < nst;
        const uint32_t cnt = on ? s_cnt[sidx] : 0u;
        const uint32_t nch = (cnt + chunk - 1u) / chunk;
        uint32_t ktot;
        const uint32_t kin = block_incl_scan<16>(nch, s_wave, &ktot);
        const uint32_t k0 = kcarry + kin - nch;
        if (j < 8 * per && j - x * per == 0) hier[8 + x] = k0;     // first key of class x
        if (on) {
            chunk_first[sidx] = k0;
            const uint32_t c0 = s_first[sidx];
            for (uint32_t k = 0; k < nch; ++k)
                desc[k0 + k] = make_uint4((uint32_t)sidx, c0 + k * chunk, min(c0 + cnt, c0 + (k + 1u) * chunk), k);
        }
        kcarry += ktot;
    }
    if (threadIdx.x == 0) { coarse_first[nst] = ccarry; hier[1] = kcarry; hier[16] = kcarry; }
}

// A wave's share of a chunk: elements [wb, we).  Unpacked rectangle of an element word (above bit kb).
__device__ __forceinline__ void hier_unpack(uint32_t word, int kb, uint32_t& lx0, uint32_t& ly0, uint32_t& lw, uint32_t& lh) {
    const uint32_t r = word >> kb;
    lx0 = r & 7u; ly0 = (r >> 3) & 7u; lw = ((r >> 6) & 7u) + 1u; lh = ((r >> 9) & 7u) + 1u;
}

// Which of a wave's 64 elements (one per lane; `valid`) cover tile `lane` = 8 * row + column of the super-tile: a 64-bit
// mask per lane, bit j = element j.  An element covers the rows [ly0, ly0 + lh) and the columns [lx0, lx0 + lw), so the
// 64 x 64 bit matrix is the AND of eight row ballots and eight column ballots: 16 ballots, parked in lanes 0..15 of a
// register pair and fetched by tile.
__device__ __forceinline__ uint64_t hier_cover(uint32_t word, int kb, bool valid, int lane) {
    uint32_t lx0, ly0, lw, lh;
    hier_unpack(word, kb, lx0, ly0, lw, lh);
    if (!valid) { lw = 0u; lh = 0u; }
    int vlo = 0, vhi = 0;
    // (v_writelane_b32 by inline assembly: this compiler has no builtin for it; the lane is an inline constant.  gfx940+: a
    // VALU instruction that reads an SGPR written by a VALU instruction needs two wait states in between, which the
    // compiler provides for its own instructions but cannot for the inside of an asm statement -- hence the s_nop 1: without
    // it the column ballots arrived one v_cmp late)
#define HS_COVER_Q(q)                                                                                                    \
    {                                                                                                                    \
        const uint64_t r = __ballot((uint32_t)(q) - ly0 < lh); /* (unsigned: ly0 <= q < ly0 + lh) */                     \
        const uint64_t c = __ballot((uint32_t)(q) - lx0 < lw);                                                           \
        asm volatile("s_nop 1\n\tv_writelane_b32 %0, %2, " #q "\n\tv_writelane_b32 %1, %3, " #q                                     \
                     : "+v"(vlo), "+v"(vhi) : "s"((uint32_t)r), "s"((uint32_t)(r >> 32)));                               \
        asm volatile("s_nop 1\n\tv_writelane_b32 %0, %2, 8+" #q "\n\tv_writelane_b32 %1, %3, 8+" #q                                 \
                     : "+v"(vlo), "+v"(vhi) : "s"((uint32_t)c), "s"((uint32_t)(c >> 32)));                               \
    }
    HS_COVER_Q(0) HS_COVER_Q(1) HS_COVER_Q(2) HS_COVER_Q(3) HS_COVER_Q(4) HS_COVER_Q(5) HS_COVER_Q(6) HS_COVER_Q(7)
#undef HS_COVER_Q
    const int ra = (lane >> 3) * 4, ca = (8 + (lane & 7)) * 4;
    const uint32_t rlo = (uint32_t)__builtin_amdgcn_ds_bpermute(ra, vlo), rhi = (uint32_t)__builtin_amdgcn_ds_bpermute(ra, vhi);
    const uint32_t clo = (uint32_t)__builtin_amdgcn_ds_bpermute(ca, vlo), chi = (uint32_t)__builtin_amdgcn_ds_bpermute(ca, vhi);
    return ((uint64_t)(rhi & chi) << 32) | (uint64_t)(rlo & clo);
}

__global__ void __launch_bounds__(256) hier_count_kernel(const uint32_t* hier, const uint4* desc, const uint2* sorted, int kb,
                                                         uint2* counts, uint32_t* tile_total, uint32_t wave_elems) {
    __shared__ uint32_t s_cnt[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int xcd = blockIdx.x % 8;
    const uint32_t c_end = hier[8 + xcd + 1];
    for (uint32_t c = hier[8 + xcd] + blockIdx.x / 8; c < c_end; c += gridDim.x / 8) {
        const uint4 d = desc[c];
        const uint32_t wb = min(d.z, d.y + (uint32_t)wave * wave_elems), we = min(d.z, wb + wave_elems);
        uint32_t w4[kHierRoundsMax];
#pragma unroll
        for (int r = 0; r < kHierRoundsMax; ++r) {   // (all loads in flight)
            const uint32_t idx = wb + r * 64 + lane;
            w4[r] = idx < we ? sorted[idx].x : 0u;
        }
        uint32_t cnt = 0;   // pairs of tile `lane` among this wave's elements
#pragma unroll
        for (int r = 0; r < kHierRoundsMax; ++r) {
            if (wb + r * 64 >= we) break;
            cnt += (uint32_t)__popcll(hier_cover(w4[r], kb, wb + r * 64 + lane < we, lane));
        }
        s_cnt[wave][lane] = cnt;
        __syncthreads();
        if (threadIdx.x < 64) {
            const uint32_t c0 = s_cnt[0][lane], c1 = s_cnt[1][lane], c2 = s_cnt[2][lane], c3 = s_cnt[3][lane];
            counts[(int64_t)c * 64 + lane] = make_uint2(c0 | (c1 << 16), c2 | (c3 << 16));
            const uint32_t tot = (c0 + c1) + (c2 + c3);
            if (tot) atomicAdd(&tile_total[(int64_t)d.x * 64 + lane], tot);
        }
        __syncthreads();
    }
}

// ranges = exclusive scan of the tile totals in tile-id order (pose, row, column); tiles without pairs keep the (0, 0) they
// were cleared to.  tile_start[(pose, super-tile) key * 64 + tile inside it] = first sorted position.  One workgroup per
// 1024 tiles (BASELINE c3: eight; one workgroup per 8192 took 10.5 us there against 6.1: the launch is all latency, and a
// thread with one tile has one load and one store on its critical path); a workgroup adds up the totals in front of its tiles
// itself (c4: 65 280 tiles, 64 workgroups, at most 63 gathered loads per thread) -- no chain, no second kernel.
#ifndef HS_TUNE_HIER_TILES_E
#define HS_TUNE_HIER_TILES_E 1   // tiles per thread (measured, kernel us at c3 / c4: 8 -> 10.5 / 28.4, 2 -> 6.6 / 24.3, 1 -> 6.1 / 23.5)
#endif
constexpr int kHierTilesPerThread = HS_TUNE_HIER_TILES_E, kHierTilesPerBlock = 1024 * kHierTilesPerThread;
__device__ __forceinline__ uint32_t hier_tile_slot(uint32_t t, uint32_t tpp, uint32_t gx, int sgx, int sgy) {
    const uint32_t pose = t / tpp, rem = t - pose * tpp, ty = rem / gx, tx = rem - ty * gx;
    return (pose * (uint32_t)(sgx * sgy) + (ty / kSuper) * (uint32_t)sgx + tx / kSuper) * 64u + (ty % kSuper) * 8u + tx % kSuper;
}
__global__ void __launch_bounds__(1024) hier_tiles_kernel(int gx, int gy, int n_poses, const uint32_t* tile_total,
                                                          uint32_t* tile_start, uint2* ranges) {
    __shared__ uint32_t s_wave[16];
    // (an empty or overflowed frame has no chunks, hence no pairs in any tile: nothing is written, the ranges stay cleared)
    const int sgx = (gx + kSuper - 1) / kSuper, sgy = (gy + kSuper - 1) / kSuper;
    const uint32_t vtiles = (uint32_t)(gx * gy * n_poses);   // (< 2^31: hs_plan)
    constexpr int E = kHierTilesPerThread;
    const uint32_t tpp = (uint32_t)(gx * gy);
    const uint32_t base = blockIdx.x * (uint32_t)kHierTilesPerBlock;
    uint32_t carry = 0;
    if (blockIdx.x > 0) {   // pairs of all tiles in front of this workgroup's
        uint32_t part = 0;
        for (uint32_t t = threadIdx.x; t < base; t += 1024) part += tile_total[hier_tile_slot(t, tpp, (uint32_t)gx, sgx, sgy)];
        uint32_t all;
        block_incl_scan<16>(part, s_wave, &all);
        carry = all;
    }
    uint32_t v[E], at[E], sum = 0;
    // (pose, row, column) of the thread's first tile by division, of the others by carrying
    const uint32_t t0 = base + threadIdx.x * E;
    uint32_t pose = t0 / tpp, rem = t0 - pose * tpp, ty = rem / (uint32_t)gx, tx = rem - ty * (uint32_t)gx;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        v[k] = 0u; at[k] = 0u;
        if (t0 + k < vtiles) {
            at[k] = (pose * (uint32_t)(sgx * sgy) + (ty / kSuper) * (uint32_t)sgx + tx / kSuper) * 64u + (ty % kSuper) * 8u + tx % kSuper;
            v[k] = tile_total[at[k]];
        }
        sum += v[k];
        if (++tx == (uint32_t)gx) { tx = 0u; if (++ty == (uint32_t)gy) { ty = 0u; ++pose; } }
    }
    uint32_t total;
    const uint32_t incl = block_incl_scan<16>(sum, s_wave, &total);
    uint32_t run = carry + incl - sum;
#pragma unroll
    for (int k = 0; k < E; ++k) {
        if (t0 + k < vtiles) {
            tile_start[at[k]] = run;
            if (v[k]) ranges[t0 + k] = make_uint2(run, run + v[k]);
        }
        run += v[k];
    }
}

__global__ void __launch_bounds__(256) hier_scatter_kernel(const uint32_t* hier, const uint4* desc, const uint2* sorted, int kb,
                                                           const uint2* counts, const uint32_t* chunk_first,
                                                           const uint32_t* tile_start, uint32_t* point_list,
                                                           const hs_counters* counters, uint32_t* counters_host,
                                                           uint32_t wave_elems) {
    __shared__ uint32_t s_part[4][64];
    __shared__ uint32_t s_inst[4][64];
    __shared__ uint2 s_stage[4][kHierStage];      // (instance, destination) of a round's pairs, tile after tile
    if (counters_host && blockIdx.x == 0 && threadIdx.x < 8)   // (the stage's last kernel: see tile_ranges_kernel)
        __hip_atomic_store(counters_host + threadIdx.x, reinterpret_cast<const uint32_t*>(counters)[threadIdx.x],
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int xcd = blockIdx.x % 8;
    const uint32_t c_end = hier[8 + xcd + 1];
    for (uint32_t c = hier[8 + xcd] + blockIdx.x / 8; c < c_end; c += gridDim.x / 8) {
        const uint4 d = desc[c];
        const uint32_t wb = min(d.z, d.y + (uint32_t)wave * wave_elems), we = min(d.z, wb + wave_elems);
        uint2 e4[kHierRoundsMax];
#pragma unroll
        for (int r = 0; r < kHierRoundsMax; ++r) {   // (this wave's elements: all loads in flight under the sums below)
            const uint32_t idx = wb + r * 64 + lane;
            e4[r] = idx < we ? sorted[idx] : make_uint2(0u, 0u);
        }
        {   // pairs of tile `lane` in the earlier chunks of this super-tile: wave w adds rows w, w + 4, ...
            const int64_t c0 = (int64_t)c - d.w;          // (= chunk_first[key])
            uint32_t part = 0;
            for (int64_t q = c0 + wave; q < c; q += 4) {
                const uint2 v = counts[q * 64 + lane];
                part += (v.x & 0xFFFFu) + (v.x >> 16) + (v.y & 0xFFFFu) + (v.y >> 16);
            }
            s_part[wave][lane] = part;
        }
        __syncthreads();
        // next position of tile `lane` for this wave: start of the tile + earlier chunks + earlier waves of this chunk
        uint32_t pos;
        {
            const uint2 v = counts[(int64_t)c * 64 + lane];
            pos = tile_start[(int64_t)d.x * 64 + lane] + (s_part[0][lane] + s_part[1][lane]) + (s_part[2][lane] + s_part[3][lane]);
            if (wave > 0) pos += v.x & 0xFFFFu;
            if (wave > 1) pos += v.x >> 16;
            if (wave > 2) pos += v.y & 0xFFFFu;
        }
        // 64 elements at a time: lane = tile walks the elements that cover it, in element order (= depth order).  The pairs
        // go through LDS -- tile after tile, each with its destination -- and leave it slot by slot, so that neighbouring lanes
        // store to neighbouring addresses (a tile's run): 64 lanes each storing into its own tile's stream cost 22 of the
        // kernel's 39 us at c3, whatever the addresses (measured with all stores folded into 256 KB: the cost is per
        // uncoalesced lane, not per byte).  A round with more pairs than the buffer holds (large Gaussians) stores directly.
#pragma unroll
        for (int r = 0; r < kHierRoundsMax; ++r) {
            if (wb + r * 64 >= we) break;
            uint64_t cover = hier_cover(e4[r].x, kb, wb + r * 64 + lane < we, lane);
            s_inst[wave][lane] = e4[r].y;      // (wave-private rows, in-order LDS: no barriers)
            const uint32_t m = (uint32_t)__popcll(cover);
            const uint32_t incl = wave_incl_scan(m, lane);
            const uint32_t total = __shfl(incl, 63);
            if (HS_TUNE_HIER_STAGE && total <= (uint32_t)kHierStage) {
                uint32_t o = incl - m, dst = pos;
                while (cover) {
                    const int j = __builtin_ctzll(cover);
                    cover &= cover - 1ull;
                    s_stage[wave][o++] = make_uint2(s_inst[wave][j], dst++);
                }
                pos = dst;
                for (uint32_t q = lane; q < total; q += 64) {
                    const uint2 v = s_stage[wave][q];
                    point_list[v.y] = v.x;
                }
            } else {
                while (cover) {
                    const int j = __builtin_ctzll(cover);
                    cover &= cover - 1ull;
                    point_list[pos++] = s_inst[wave][j];
                }
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------- tile ranges (a8)
// Four consecutive sorted tile ids per thread (one 16-byte load; the id before the first comes from the neighbouring
// lane): ranges[t] = [first, last + 1) of tile t's run; tiles without pairs keep the (0, 0) they were cleared to.
// (the stage's last kernel: its first thread also leaves a copy of the frame's counters at `counters_host`, hs_fwd_args --
// every earlier kernel of the stage has finished, so num_rendered, the overflow verdict and the help count are final)
__global__ void __launch_bounds__(256) tile_ranges_kernel(const uint32_t* tiles, const uint32_t* n_dev, uint2* ranges,
                                                          const hs_counters* counters, uint32_t* counters_host) {
    if (counters_host && blockIdx.x == 0 && threadIdx.x < 8)
        __hip_atomic_store(counters_host + threadIdx.x, reinterpret_cast<const uint32_t*>(counters)[threadIdx.x],
                           __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    const int64_t n = *n_dev;
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    const int lane = threadIdx.x & 63;
    uint32_t t[4] = {0u, 0u, 0u, 0u};
    if (i0 + 4 <= n) {
        const uint4 q = *reinterpret_cast<const uint4*>(tiles + i0);
        t[0] = q.x; t[1] = q.y; t[2] = q.z; t[3] = q.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) t[k] = i0 + k < n ? tiles[i0 + k] : 0u;
    }
    uint32_t prev = __shfl_up(t[3], 1);
    if (lane == 0 && i0 > 0 && i0 < n) prev = tiles[i0 - 1];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int64_t i = i0 + k;
        if (i < n) {
            if (i == 0) ranges[t[k]].x = 0;
            else if (t[k] != prev) { ranges[prev].y = (uint32_t)i; ranges[t[k]].x = (uint32_t)i; }
            if (i == n - 1) ranges[t[k]].y = (uint32_t)n;
        }
        prev = t[k];
    }
}

// P == 0: one small kernel instead of the memset + copy the call used to enqueue (a captured step holds kernels only)
__global__ void __launch_bounds__(256) empty_frame_kernel(hs_counters* clear, uint2* ranges, int64_t ntiles, uint32_t* counters_host,
                                                          const hs_counters* counters) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (ranges && i < ntiles) ranges[i] = make_uint2(0u, 0u);
    if (blockIdx.x == 0 && threadIdx.x < 8) {
        const uint32_t v = clear ? 0u : reinterpret_cast<const uint32_t*>(counters)[threadIdx.x];
        if (clear) reinterpret_cast<uint32_t*>(clear)[threadIdx.x] = 0u;
        if (counters_host) __hip_atomic_store(counters_host + threadIdx.x, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

}  // namespace

int launch_empty_frame(hs_counters* clear, uint2* ranges, int64_t ntiles, uint32_t* counters_host, const hs_counters* counters,
                       hipStream_t s) {
    if (!clear && !ranges && !counters_host) return HS_OK;
    empty_frame_kernel<<<ranges ? ceil_div(ntiles, 256) : 1, 256, 0, s>>>(clear, ranges, ntiles, counters_host, counters);
    HS_LAUNCH_CHECK();
    return HS_OK;
}

int64_t sort_tmp_bytes(int64_t n) {
    // every sort uses the same tile; 8 passes of 256-bin status words (hs_sort_pairs on 64-bit keys) is also what the depth
    // sort's 4 passes of 512-bin words need
    static_assert(8 * 256 >= 4 * kDepthBins && kDepthTile >= kSortTileMin, "sort_tmp_bytes covers the depth sort's status words");
    return align_up(sort_scratch_words(n > 0 ? n : 1, 8, kSortTileMin) * 4, 256);
}

// hs_sort_pairs: stable LSD radix sort of (u64 key, u32 value) pairs with the same pass kernel (keys and values in
// separate arrays).  Ping-pongs between (k0,v0) and (k1,v1); the result lands in (k0,v0) when sort_passes(nbits) is
// even, else in (k1,v1).  `fail_word`: device word that reads 2 afterwards if a look-back gave up.
int launch_radix_sort(uint64_t* k0, uint32_t* v0, uint64_t* k1, uint32_t* v1, const uint32_t* n_dev,
                      int64_t n_launch, int nbits, void* tmp, uint32_t* fail_word, hipStream_t s) {
    if (n_launch <= 0) return HS_OK;
    constexpr int ITEMS = kU64SortItems, TILE = ITEMS * kSortBlock;
    const int nblk = ceil_div(n_launch, TILE);
    const int passes = sort_passes(nbits);
    const SortScratch sc(tmp, n_launch, TILE);
    HS_HIP_CHECK(hipMemsetAsync(tmp, 0, (size_t)sort_scratch_words(n_launch, passes, TILE) * 4, s));
    if (fault_injection() == 1)   // tests only (HS_FAULT_INJECT=sort_ticket): chain position 0 is never handed out
        HS_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)sc.tickets, 1, 1, s));
    radix_ghist_kernel<uint64_t, false><<<ceil_div(n_launch, kHistTile), kHistThreads, 0, s>>>(k0, n_dev, nbits, passes, sc.ghist);
    uint64_t* kin = k0; uint32_t* vin = v0; uint64_t* kout = k1; uint32_t* vout = v1;
    int pass = 0;
    for (int shift = 0, w = 0; shift < nbits; shift += w, ++pass) {
        w = (nbits - shift + (passes - pass) - 1) / (passes - pass);
        // (always the TICKET instantiation: it is the one the fault injection of the tests can reach, and this entry point
        // is not on the pipeline's path)
        radix_sweep_kernel<uint64_t, ITEMS, 8, false, false, true><<<nblk, kSortBlock, 0, s>>>(
            kin, vin, kout, vout, n_dev, shift, (1u << w) - 1u, sc.status + (int64_t)pass * sc.pass_words,
            sc.ghist + 256 * pass, sc.tickets + pass, fail_word, nullptr);
        HS_LAUNCH_CHECK();
        uint64_t* tk = kin; kin = kout; kout = tk;
        uint32_t* tv = vin; vin = vout; vout = tv;
    }
    return HS_OK;
}

// a5 in instance order, as the published pipeline lays it out: offsets[i] = sum_{j<=i} tiles_touched[j], total =
// num_rendered.  Run by PREPROCESS-only calls (the upstream-style host read of num_rendered before binning) and by
// HS_STAGE_OFFSETS (inspection); a single-enqueue forward does not need it -- the binning stage scans the
// depth-ordered counts, whose total is the same R.
int launch_scan(const hs_fwd_args& a, const hs_layout& L, hipStream_t s) {
    const hs_dims& d = a.dims;
    char* geom = (char*)a.geom;
    const int64_t I = (int64_t)d.P * d.n_poses;
    hs_counters* counters = (hs_counters*)(geom + L.counters);
    return scan_u32((const uint32_t*)(geom + L.tiles_touched), I, (uint32_t*)(geom + L.scan_spine),
                    (uint32_t*)(geom + L.offsets), &counters->num_rendered, s);
}

int launch_tile_keys(const hs_fwd_args& a, const hs_layout& L, hipStream_t s) {
    const hs_dims& d = a.dims;
    const int gx = (d.W + kTile - 1) / kTile, gy = (d.H + kTile - 1) / kTile;
    const int vtiles = gx * gy * d.n_poses;
    char* bin = (char*)a.binning;
    tile_keys_kernel<<<ceil_div(vtiles, 4), 256, 0, s>>>((const uint2*)(bin + L.ranges), vtiles, (uint32_t)d.capacity,
                                                        (uint32_t*)(bin + L.keys_sorted),
                                                        (const hs_counters*)((const char*)a.geom + L.counters));
    HS_LAUNCH_CHECK();
    return HS_OK;
}

int launch_binning(const hs_fwd_args& a, const hs_layout& L, hipStream_t s, uint32_t frame_tag) {
    const hs_dims& d = a.dims;
    char* geom = (char*)a.geom;
    char* bin = (char*)a.binning;
    const int64_t I = (int64_t)d.P * d.n_poses;
    const int gx = (d.W + kTile - 1) / kTile, gy = (d.H + kTile - 1) / kTile;
    const int64_t ntiles = (int64_t)gx * gy * d.n_poses;
    hs_counters* counters = (hs_counters*)(geom + L.counters);
    uint32_t* n_sort = &counters->reserved[0];
    const uint32_t* n_inst = &counters->reserved[1];
    uint2* ranges = (uint2*)(bin + L.ranges);
    // when preprocess ran in this same call it already wrote the depth keys, the instance count, cleared the ranges and
    // both scratch regions
    const bool prepared = (a.stages & HS_STAGE_PREPROCESS) != 0;
    void* tmp = bin + L.sort_tmp;                         // scratch of the depth sort
    uint64_t* scan_status = (uint64_t*)(bin + L.pair_sort_tmp);   // one word per emission block, then the pair sort's scratch
    void* tmp2 = (char*)scan_status + emit_scan_words(I) * 4;
    const int tbits = tile_bits((uint32_t)ntiles);
    const int passes = sort_passes(tbits);
    const SortScratch dsc(tmp, I, kDepthTile, kDepthBins);
    unsigned long long* depth_bits = reinterpret_cast<unsigned long long*>(dsc.tickets + kDepthBitsAt);
    const uint32_t dtag = prepared ? frame_tag : 0u;   // (not prepared: bin_prepare_kernel zeroes the words, tag 0)
    if (!prepared) {
        // (a grid wide enough for the scratch it clears: a few MB of status words at several million instances)
        const int64_t nz = depth_scratch_words(I) + pair_scratch_words(I, d.capacity, passes);
        const int grid = (int)max((int64_t)ceil_div(ntiles, 256), min((int64_t)4096, (nz + 4 * 256 - 1) / (4 * 256)));
        bin_prepare_kernel<<<grid, 256, 0, s>>>(counters, (uint32_t)I, ranges, ntiles, (uint32_t*)tmp, depth_scratch_words(I),
                                                (uint32_t*)scan_status, pair_scratch_words(I, d.capacity, passes));
    }

    if (fault_injection() == 2 && !sort_tickets())   // tests only (HS_FAULT_INJECT=stalled_chain): the verdict of a stalled chain
        HS_HIP_CHECK(hipMemsetD32Async((hipDeviceptr_t)&counters->overflow, 2, 1, s));
    // 1. instances by depth: stable, over the bits of the depth keys that VARY, in digits of up to nine bits (three passes
    //    for a scene spanning up to 2^4 in depth; the layout lives on the device, so four passes are enqueued and those
    //    without a digit return at once); (key, instance) elements, the last pass leaves only the instance list
    uint2* dp0 = (uint2*)(bin + L.depth_pairs);
    uint2* dp1 = dp0 + I;
    uint32_t* inst_sorted = (uint32_t*)(bin + L.inst_sorted);
    if (!prepared)
        depth_keys_kernel<<<ceil_div(I, 1024), 256, 0, s>>>(I, (const float*)(geom + L.depth), (const int*)(geom + L.radii), dp0,
                                                            depth_bits);
    const int mode = tile_sort_mode(I, gx, gy, d.n_poses, d.capacity);
    const int eblk = ceil_div(I, 256);
    // The rectangles in depth order and the sums per 256 instances of that order, which the emission's offsets come from:
    // written by the counting depth sort's last kernel on its way out, else by gather_binfo_kernel / hier_gather_kernel
    bool gathered = false;
    uint32_t* bsum = (uint32_t*)(dp0 + I);   // (= dp1: free once the depth sort is done)
    if (depth_sort_mode(I) == kDepthSortMsd) {
        // frames below 2^21 instances: one stable counting pass over the top varying bits + range sorts in LDS (kernels above)
        const int rows = (int)depth_msd_rows(I);
        uint32_t* dw = (uint32_t*)(bin + L.depth_ws);
        uint32_t* counts2 = dw;                                             // [rows][2048] u16 pairs
        uint32_t* bases = counts2 + (int64_t)rows * (kMsdBuckets / 2);      // [rows][4096]
        uint32_t* totals = bases + (int64_t)rows * kMsdBuckets;             // [4096]
        uint32_t* culled_rows = totals + kMsdBuckets;                       // [rows]
        depth_msd_count_kernel<4><<<rows, depth_msd_tile(I) / 4, 0, s>>>(dp0, n_inst, depth_bits, dtag, counts2, culled_rows);
        depth_msd_colscan_kernel<<<kMsdBuckets / 32, 1024, 0, s>>>(counts2, rows, (uint2*)bases, (uint2*)totals);
        // the emission of a frame this size takes its rectangles gathered (unless HS_SCAN_IN_EMISSION=1 asks it to gather
        // them itself): the range sort does that on its way out, the block sums live in the count rows (dead by then)
        GatherOut G = {nullptr, nullptr, nullptr, nullptr, (uint32_t)I};
        uint32_t* zero_b = nullptr;
        int64_t n_zero_b = 0;
        if (HS_TUNE_FUSE_GATHER && (mode == kTileSortHier || !scan_in_emission(I))) {
            gathered = true;
            bsum = counts2;
            G.binfo = (const uint2*)(geom + L.binfo); G.srect = dp0; G.bsum = bsum;
            if (mode == kTileSortHier) {
                G.bcsum = bsum + eblk;
                zero_b = (uint32_t*)(bin + L.hier_ws);
                n_zero_b = HierWs(gx, gy, d.n_poses, d.capacity).zero_words;
            }
        }
        const int64_t n_zero_a = gathered ? 2 * (int64_t)eblk : 0;
        if (depth_msd_tile(I) == 1024)
            depth_msd_scatter_kernel<256, 4><<<rows, 256, 0, s>>>(dp0, n_inst, depth_bits, dtag, bases, totals, culled_rows, dp1, inst_sorted,
                                                                  bsum, n_zero_a, zero_b, n_zero_b);
        else
            depth_msd_scatter_kernel<512, 8><<<rows, 512, 0, s>>>(dp0, n_inst, depth_bits, dtag, bases, totals, culled_rows, dp1, inst_sorted,
                                                                  bsum, n_zero_a, zero_b, n_zero_b);
        const int range = depth_msd_range(I);
        depth_range_sort_kernel<<<ceil_div(I, range) + 1, kRangeThreads, 0, s>>>(dp1, dp0, depth_bits, dtag, totals, inst_sorted,
                                                                                 counters, depth_range_cap(),
                                                                                 max(1, tile_bits((uint32_t)(I - 1))), depth_dist_max(), G, range);
        HS_LAUNCH_CHECK();
    } else {
        const int nblk = ceil_div(I, kDepthTile);
        depth_ghist_kernel<<<ceil_div(I, 4096), 1024, 0, s>>>(dp0, n_inst, depth_bits, dtag, dsc.ghist);
        const uint32_t late = fault_injection() == 3 ? 0x80000000u : 0u;
        for (int pass = 0; pass < 4; ++pass) {
            uint32_t* st = dsc.status + (int64_t)pass * dsc.pass_words;
            if (sort_tickets())
                radix_sweep_kernel<uint32_t, kDepthSortItems, kDepthSortLook, true, true, true, kDepthBins, true>
                    <<<nblk, kDepthBins, 0, s>>>(dp0, nullptr, dp1, inst_sorted, n_inst, 0, late, st, dsc.ghist, dsc.tickets + pass,
                                                 &counters->overflow, nullptr, depth_bits, pass, dtag);
            else
                radix_sweep_kernel<uint32_t, kDepthSortItems, kDepthSortLook, true, true, false, kDepthBins, true>
                    <<<nblk, kDepthBins, 0, s>>>(dp0, nullptr, dp1, inst_sorted, n_inst, 0, late, st, dsc.ghist, dsc.tickets + pass,
                                                 &counters->overflow, nullptr, depth_bits, pass, dtag);
        }
        HS_LAUNCH_CHECK();
    }
    // 2. pair emission in depth order, with the scan of the pair counts inside (emit_pairs_kernel); it also counts the
    //    digit totals of the tile sort below.  The last pass of that sort writes (keys_sorted, point_list), which overlay
    //    packed buffer A: it must read buffer B, so the emission starts in A when the pass count is even
    uint32_t* offs = (uint32_t*)(bin + L.offs_sorted);   // inclusive pair offsets in depth order (the backward's segmented sum walks them)
    uint2* pA = (uint2*)(bin + L.keys_sorted);
    uint2* pB = (uint2*)(bin + L.pairs_tmp);
    // small frames: counting sort by tile id (the kernels above); the emission then always writes buffer B
    if (mode == kTileSortHier) {
        // hierarchical form (kernels above).  Elements start in packed buffer A when the pass count is odd, so that the
        // sorted list ends in B: A's second half is point_list, which the expansion writes while it reads the list
        const HierWs W(gx, gy, d.n_poses, d.capacity);
        uint32_t* hw = (uint32_t*)(bin + L.hier_ws);
        const int kb = tile_bits((uint32_t)W.nst), cpasses = sort_passes(kb);
        uint2* srect = dp0;
        uint32_t* bcsum = bsum + eblk;
        const bool big = I >= (2 << 20);
        uint32_t* ssum = big ? bcsum + eblk : nullptr;     // (big frames: sums of 1024 instances too, ceil(I / 1024) words each)
        uint32_t* scsum = big ? ssum + ceil_div(I, 1024) : nullptr;
        if (gathered) {
            // (the counting depth sort left the rectangles, the sums and the cleared words)
        } else if (big)
            hier_gather_kernel<1024><<<ceil_div(I, 1024), 1024, 0, s>>>(I, inst_sorted, (const uint2*)(geom + L.binfo), srect, counters,
                                                                        bsum, bcsum, ssum, scsum, hw, W.zero_words);
        else
            hier_gather_kernel<256><<<eblk, 256, 0, s>>>(I, inst_sorted, (const uint2*)(geom + L.binfo), srect, counters, bsum, bcsum,
                                                         nullptr, nullptr, hw, W.zero_words);
        // (beyond 32768 emission workgroups -- 8 M instances -- a scan kernel turns the 256-instance sums into prefixes)
        const bool excl_ready = eblk > 4 * 8192;
        if (excl_ready) {
            scan_spine_kernel<<<1, 256, 0, s>>>(bsum, eblk, nullptr);
            scan_spine_kernel<<<1, 256, 0, s>>>(bcsum, eblk, nullptr);
        }
        uint2* e0 = cpasses % 2 != 0 ? pA : pB;
        uint2* e1 = e0 == pA ? pB : pA;
        const SortScratch sc(tmp2, d.capacity, kPairTile);
#define HS_HEMIT_ARGS I, d.P, gx, gy, (float4*)(geom + L.rec), inst_sorted, srect, bsum, bcsum, ssum, scsum, offs, e0,                \
                      (uint8_t*)(bin + L.pair_flags), counters, (uint64_t)d.capacity, sc.ghist, kb, cpasses, depth_bits,             \
                      (int)excl_ready, hw, hw + W.st_count, (int)W.nst, (int)W.nst_pad()
        if (big) hier_emit_kernel<true><<<eblk, 256, 0, s>>>(HS_HEMIT_ARGS);
        else hier_emit_kernel<false><<<eblk, 256, 0, s>>>(HS_HEMIT_ARGS);
#undef HS_HEMIT_ARGS
        HS_LAUNCH_CHECK();
        {
            const int nblk = ceil_div(d.capacity, kPairTile);
            uint2* in = e0; uint2* out = e1;
            int pass = 0;
            for (int shift = 0, w = 0; shift < kb; shift += w, ++pass) {
                w = (kb - shift + (cpasses - pass) - 1) / (cpasses - pass);
                uint32_t* st = sc.status + (int64_t)pass * sc.pass_words;
                const uint32_t mask = ((1u << w) - 1u) | (fault_injection() == 3 ? 0x80000000u : 0u);
                if (sort_tickets())
                    radix_sweep_kernel<uint32_t, kPairSortItems, kPairSortLook, true, true, true><<<nblk, kSortBlock, 0, s>>>(
                        in, nullptr, out, nullptr, hw, shift, mask, st, sc.ghist + 256 * pass, sc.tickets + pass,
                        &counters->overflow, hw);
                else
                    radix_sweep_kernel<uint32_t, kPairSortItems, kPairSortLook, true, true, false><<<nblk, kSortBlock, 0, s>>>(
                        in, nullptr, out, nullptr, hw, shift, mask, st, sc.ghist + 256 * pass, sc.tickets + pass,
                        &counters->overflow, hw);
                HS_LAUNCH_CHECK();
                uint2* t = in; in = out; out = t;
            }
            // `in` = the sorted elements (packed buffer B)
            const uint32_t chunk = (uint32_t)hier_chunk(I);
            hier_plan_kernel<<<1, 1024, 0, s>>>(counters, hw, hw + W.st_count, (int)W.nst, (int)W.nst_pad(), hw + W.coarse_first,
                                                hw + W.chunk_first, (uint4*)(hw + W.desc), chunk);
            // expansion workgroups: 8 XCDs x kHierGridPerXcd, each walking its XCD's chunks
            const int egrid = 8 * (int)min((int64_t)kHierGridPerXcd, (W.chunks_max + 7) / 8);
            hier_count_kernel<<<egrid, 256, 0, s>>>(hw, (const uint4*)(hw + W.desc), in, kb, (uint2*)(hw + W.counts),
                                                    hw + W.tile_total, chunk / 4u);
            hier_tiles_kernel<<<ceil_div(ntiles, kHierTilesPerBlock), 1024, 0, s>>>(gx, gy, d.n_poses, hw + W.tile_total,
                                                                                    hw + W.tile_start, ranges);
            hier_scatter_kernel<<<egrid, 256, 0, s>>>(hw, (const uint4*)(hw + W.desc), in, kb,
                                                                  (const uint2*)(hw + W.counts), hw + W.chunk_first,
                                                                  hw + W.tile_start, (uint32_t*)(bin + L.point_list), counters,
                                                                  (uint32_t*)a.counters_host, chunk / 4u);
            HS_LAUNCH_CHECK();
        }
        return HS_OK;
    }
    const bool counting = mode == kTileSortCount;
    uint2* p0 = (counting || passes % 2 != 0) ? pB : pA;
    uint2* p1 = p0 == pA ? pB : pA;
    uint32_t* t_counts = (uint32_t*)(bin + L.tile_matrix);            // [emission workgroup][key]: pairs, one byte per wave
    uint32_t* t_bases = t_counts + (int64_t)ceil_div(I, 256) * ntiles; // ... pairs of that key in earlier workgroups
    uint32_t* t_totals = t_bases + (int64_t)ceil_div(I, 256) * ntiles; // [key]
    // (the depth sort's two buffers are free again: they hold the gathered rectangles and their block sums when the
    // offsets are computed ahead of the emission)
    uint2* srect = nullptr;
    uint32_t* block_excl = nullptr;
    bool excl_ready = false;
    if (!scan_in_emission(I)) {
        srect = dp0;
        block_excl = bsum;
        if (!gathered)
            gather_binfo_kernel<<<eblk, 256, 0, s>>>(I, inst_sorted, (const uint2*)(geom + L.binfo), srect, counters, block_excl);
        excl_ready = eblk > 8192;
        if (excl_ready) scan_spine_kernel<<<1, 256, 0, s>>>(block_excl, eblk, nullptr);
    }
#define HS_EMIT_ARGS I, d.P, gx, gy, (float4*)(geom + L.rec), inst_sorted, (const uint2*)(geom + L.binfo), srect, block_excl,       \
                     scan_status, offs, p0, (uint8_t*)(bin + L.pair_flags), counters, (uint64_t)d.capacity, (uint32_t*)tmp2, tbits, \
                     passes, depth_bits, (int)excl_ready
    if (counting) {
        if (srect) emit_pairs_kernel<false, true><<<eblk, 256, 0, s>>>(HS_EMIT_ARGS, t_counts, (int)ntiles);
        else emit_pairs_kernel<true, true><<<eblk, 256, 0, s>>>(HS_EMIT_ARGS, t_counts, (int)ntiles);
    } else if (srect) emit_pairs_kernel<false><<<eblk, 256, 0, s>>>(HS_EMIT_ARGS);
    else emit_pairs_kernel<true><<<eblk, 256, 0, s>>>(HS_EMIT_ARGS);
#undef HS_EMIT_ARGS
    HS_LAUNCH_CHECK();
    if (counting) {
        tile_colscan_kernel<<<ceil_div(ntiles, kColscanCols), kColscanCols * kColscanSlots, 0, s>>>(
            t_counts, eblk, (int)ntiles, n_sort, t_bases, t_totals);
        tile_scatter_kernel<<<eblk, 256, (size_t)5 * ntiles * 4, s>>>(
            I, (int)ntiles, p0, offs, t_counts, t_bases, t_totals, n_sort,
            (uint32_t*)(bin + L.point_list), ranges, counters, (uint32_t*)a.counters_host);
        HS_LAUNCH_CHECK();
        return HS_OK;
    }
    // 3. stable sort by tile id only
    uint32_t* keys_sorted = (uint32_t*)(bin + L.keys_sorted);
    int rc = radix_sort_packed<kPairSortItems, kPairSortLook>(p0, p1, keys_sorted, (uint32_t*)(bin + L.point_list), n_sort,
                                                         d.capacity, tbits, tmp2, &counters->overflow, n_sort, s,
                                                         /*zeroed=*/true, /*ghist_ready=*/true);
    if (rc != HS_OK) return rc;
    if (d.capacity > 0) {
        tile_ranges_kernel<<<ceil_div(d.capacity, 1024), 256, 0, s>>>(keys_sorted, n_sort, ranges, counters,
                                                                      (uint32_t*)a.counters_host);
        HS_LAUNCH_CHECK();
    } else if (a.counters_host) {
        HS_HIP_CHECK(hipMemcpyAsync(a.counters_host, counters, sizeof(hs_counters), hipMemcpyDeviceToHost, s));
    }
    return HS_OK;
}

}  // namespace hs

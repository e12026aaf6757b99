// One-wave selection of the k best candidate keys (cmr_device.h: u64 keys, unsigned order = result order) — shared by the
// single-launch search of small corpora (aux_kernels.hip: tiny_search_kernel) and the finishing stage of the scan kernel
// (scan_kernels.hip: MODE_FIN).
#pragma once
#include "cmr_device.h"

// Selection by one wave over <= 16 keys per lane (0 = empty, keys unique), without rounds for k <= 64: k rounds of
// "wave-wide arg-max, remove" cost ~1300 cycles each — 13 us at k = 20.  Instead the k-th largest of the 64 per-lane maxima
// T is a lower bound of the k-th best key (k lanes hold a key >= T), so only keys >= T can win; those survivors (k .. a
// few dozen) are compacted into one key per lane and ranked by counting — lane l's key goes to output position rank(l).
// All cross-lane traffic is v_readlane.  More than 64 survivors (many lanes whose second best also beats T) or k > 64: the
// exact k-th key by a bitwise search (below).  emit(rank, key) is called exactly once for every rank < k (key 0: no such row), by one lane.
__device__ __forceinline__ u64 tiny_readlane(u64 v, int l) {      // l wave-uniform
    return ((u64)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (u64)(unsigned)__builtin_amdgcn_readlane((int)v, l);
}

template <class Emit>
__device__ __forceinline__ void tiny_select(u64 (&key)[16], int k, u64* stage, int lane, Emit emit) {
    if (k <= 64) {
        u64 best = 0ull;
#pragma unroll
        for (int j = 0; j < 16; ++j) best = key[j] > best ? key[j] : best;
        int rk = 0;
#pragma unroll
        for (int l = 0; l < 64; ++l) rk += tiny_readlane(best, l) > best ? 1 : 0;
        const u64 has = __ballot(best != 0ull && rk == k - 1);
        const u64 T = has ? tiny_readlane(best, __ffsll((long long)has) - 1) : 0ull;      // fewer than k lanes hold a key: everything survives
        int S = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const bool sv = key[j] != 0ull && key[j] >= T;
            const u64 m = __ballot(sv);
            const int slot = S + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
            if (sv && slot < 64) stage[slot] = key[j];
            S += __popcll(m);
        }
        if (S <= 64) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const u64 mine = lane < S ? stage[lane] : 0ull;
            int r = 0;
            for (int l = 0; l < S; ++l) r += tiny_readlane(mine, l) > mine ? 1 : 0;
            if (lane < S && r < k) emit(r, mine);
            if (lane >= S && lane < k) emit(lane, 0ull);          // fewer than k rows with a score: the tail is empty
            __builtin_amdgcn_wave_barrier();                         // the stage is reused by this wave's next query
            return;
        }
    }
    // k > 64 (BASELINE config 5 searches with k = 100), or more than 64 survivors above (k close to 64: the bound from the lane
    // maxima is loose): the EXACT k-th largest key by a bitwise search — for bit 63 .. 0 try prefix | bit and count the keys
    // >= it with ballots (16 compares + 16 population counts per bit, ~6 us in all) — then exactly min(k, n) survivors are
    // compacted into two keys per lane and ranked by counting.  The k rounds of "arg-max, remove" this replaces cost ~1.5 us
    // each: 160 us of a 205 us call at k = 100.
    u64 T = 0ull;
    for (int bit = 63; bit >= 0; --bit) {
        const u64 trial = T | (1ull << bit);
        int c = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) c += __popcll(__ballot(key[j] >= trial));
        if (c >= k) T = trial;               // wave-uniform
    }
    if (T == 0ull) T = 1ull;                 // fewer than k keys: every non-empty one survives
    int S = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const bool sv = key[j] >= T;
        const u64 m = __ballot(sv);
        const int slot = S + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        if (sv) stage[slot] = key[j];         // S <= k <= 128: keys are unique, exactly min(k, n) of them are >= T
        S += __popcll(m);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const u64 m0 = lane < S ? stage[lane] : 0ull, m1 = lane + 64 < S ? stage[lane + 64] : 0ull;
    int q0 = 0, q1 = 0;
    const int S0 = S < 64 ? S : 64, S1 = S - S0;
    for (int l = 0; l < S0; ++l) { const u64 o = tiny_readlane(m0, l); q0 += o > m0 ? 1 : 0; q1 += o > m1 ? 1 : 0; }
    for (int l = 0; l < S1; ++l) { const u64 o = tiny_readlane(m1, l); q0 += o > m0 ? 1 : 0; q1 += o > m1 ? 1 : 0; }
    if (m0 != 0ull) emit(q0, m0);
    if (m1 != 0ull) emit(q1, m1);
    for (int e = S + lane; e < k; e += 64) emit(e, 0ull);     // fewer than k rows with a score: the tail is empty
    __builtin_amdgcn_wave_barrier();                             // the stage is reused by this wave's next query
}

// The same selection over a STREAM of n keys (fetch(i), 0 beyond n): one chunk of <= 1024 keys, or — k <= 64 — chunks of
// 960 with the running k best carried along in the wave's LDS row (slot 15 of lanes 0 .. k-1).
// PRECONDITION: n <= 1024 or k <= 64.  The single-chunk branch looks at the first 1024 keys only, and the carry row holds 64
// keys: a stream longer than 1024 with k > 64 has no correct route here (tiny_select_stream_ok below).
// The host-side launch wrappers (tiny_launch, cmr_launch_scan_fin) refuse a launch this predicate rejects — the SAME function, so the
// guard and the invariant cannot drift apart; the device-side trap is left to development builds (a trap aborts the whole HIP context
// and every index of the process with it).
__host__ __device__ constexpr bool tiny_select_stream_ok(long long n, int k) { return n <= 1024 || k <= 64; }

template <class Fetch, class Emit>
__device__ __forceinline__ void tiny_select_stream(int n, int k, u64* stage, u64* carry, int lane, Fetch fetch, Emit emit) {
    u64 key[16];
#ifdef CMR_DEV_KNOBS
    if (__builtin_expect(!tiny_select_stream_ok(n, k), 0)) __builtin_trap();
#endif
    if (n <= 1024) {
#pragma unroll
        for (int j = 0; j < 16; ++j) key[j] = fetch(lane + 64 * j);
        tiny_select(key, k, stage, lane, emit);
        return;
    }
    for (int base = 0; base < n; base += 960) {
#pragma unroll
        for (int j = 0; j < 15; ++j) key[j] = fetch(base + lane + 64 * j);
        key[15] = (base > 0 && lane < k) ? carry[lane] : 0ull;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                 // every lane holds its carried key before the row is rewritten
        if (base + 960 >= n) tiny_select(key, k, stage, lane, emit);
        else tiny_select(key, k, stage, lane, [&](int r, u64 kk) { carry[r] = kk; });
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
}

// One-launch planner body for SMALL populations (one GPU's K/G shard of BASELINE configs[1]): the hoisted GRU encode
// (w_nl.py:14-29, one 16-window tile per WORKGROUP: gru_encode_tile_coop) and the latency-split T-step rollout (planners/mppi_delay.py:271-296,
// one 16-sample tile per workgroup) run as ROLES of the same persistent grid instead of two back-to-back launches.
//
// Why: at K = 2048 the split rollout is 128 workgroups walking 40 strictly sequential horizon steps (0.40 ms) while
// the other half of the chip idles, after the encode (0.42 ms, throughput-bound on the whole chip) has run alone.
// Here the rollout of a tile ("chain") starts once a first bank of latents exists, the encoder workgroups keep every other
// SIMD busy, and the ones beside a chain yield when the chain would otherwise fall behind (schedule: in the kernel).
//
// Grid: 4 workgroups of 256 threads per CU, all co-resident (the host sizes the grid from the device's CU count and
// this kernel's occupancy).  Roles are taken at run time and do NOT depend on dispatch order or placement for
// correctness (only for speed):
//   * census: the first workgroup to arrive on a CU (s_getreg HW_ID / XCC_ID -> per-CU counter) may take one of the first
//     roll_cap rollout tiles (census ticket + the tile's owner word), so rollout workgroups sit on distinct CUs;
//   * everybody else takes encoder tiles from the encoder ticket, in horizon-major order (all tiles of step t before
//     step t+1), one tile per workgroup (one gate chunk per wavefront), and publishes each tile's latents;
//   * when the encoder ticket runs dry the workgroup drains the rollout tiles that have no owner yet (drain ticket +
//     owner word).
// An encoder never waits for anything, so the grid drains even if a rollout workgroup had to give up (bounded spins).
//
// Hand-off of a tile's latents (256 B, (T, K, 2) horizon-major so a tile is two whole 128-B lines), following
// cdna_hip_programming.md Guideline 16 / MI355X_MICROARCH.md "inter-workgroup visibility" (third row of the sc1 table):
//   producer (wave 0 of the encoder workgroup): 8-B write-through (sc1) stores of the whole lines by ONE store instruction, s_waitcnt vmcnt(0),
//                  then ONE lane's agent-scope atomic add on the tile's flag word;
//   consumer:      ONE wave polls the flag with relaxed agent loads (global_load_dword sc1), a workgroup barrier, then
//                  EVERY load of the latents is a 16-B buffer_load ... sc1 (bypasses the CU's L1; no acquire fence).
// All polled words are zero when the launch starts (the previous command's merge kernel, the command's own perturb kernel
// when one runs, or a memset zeroes them).
//
// Round 3: for the single planner with device noise the launch is the WHOLE first phase of command():
//   * sampling / bounding (planners/mppi_delay.py:319-328): wave 0 of encoder tile (t, j) draws the actions of its 16
//     windows (Philox is counter-based: the four tiles that share an action draw the same value), stages them in LDS for
//     the workgroup and publishes perturbed / noise / actions of ITS step t with the tile's latents (same flag);
//   * U <- roll(U, -1) (:199-200) is read on the fly from the sequence before the shift; workgroup 0 stores it;
//   * the state and the action buffer are read from the kernel-argument segment (no staging copy, no perturb kernel);
//   * importance weights (:210-216): a rollout tile IS a weight tile (nlc_mppi_dev.h) -- wave 0 folds its 16 samples
//     (beta_b, eta_b, S_b) the moment their costs are final and counts the tile done; the workgroup whose count comes
//     last folds the tile partials into the shard's (beta_r, eta_r, S_r): the arithmetic of weight_tile_kernel /
//     weight_rank_kernel, bit for bit.
// command() is then this launch + merge_kernel (after the shard all-gather) instead of six launches.
#pragma once
#include "nlc_device.h"
#include "nlc_gru_tile.h"
#include "nlc_kernels.h"
#include "nlc_mppi_dev.h"
#include "nlc_rollout.h"

namespace nlc {

typedef unsigned v4u __attribute__((ext_vector_type(4)));

#define NLC_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// Diagnostics (progress counters, timeline stamps) are compiled in only with -DNLC_FUSED_TRACE=1 (tools/fused_debug.py):
// 5120 encoder tiles x 3 same-address device-scope atomics were themselves a bottleneck of the launch.
#ifndef NLC_FUSED_TRACE
#define NLC_FUSED_TRACE 0
#endif
// timeline stamps: every lane of the calling wave issues the same atomic max
__device__ __forceinline__ void stamp_max(unsigned* w, bool complement) {
  if (!NLC_FUSED_TRACE) return;
  const unsigned t = (unsigned)__builtin_amdgcn_s_memrealtime();
  __hip_atomic_fetch_max(w, complement ? ~t : t, NLC_RLX_AGENT);
}
// per-CU state word (kFusedCuState): 0 no chain, 2 the CU's chain has finished, >= kChainWalking: walking, at horizon step
// (word - kChainWalking)
constexpr unsigned kChainWalking = 16u;
// GRU latents published by encoder workgroups of this launch (see the file header for the protocol)
struct PaHandoff {
  __amdgpu_buffer_rsrc_t rsrc;  // over paT (T, K, 2)
  int64_t K;
  const unsigned* flags;  // (T, ntk)
  int ntk, tile, T;
  unsigned* sync;           // device copy of the give-up code
  unsigned* timeout_host;   // pinned host word the planner checks after the command
  double cur0, cur1, nxt0, nxt1;
  int ready_upto;  // polling wave: flags of steps < ready_upto have been seen set
  static constexpr int kPollWave = 3;        // the wave with the fewest layer-3 tiles
  unsigned spin_limit;                       // ~2 us per poll once it naps; then give up (never hang the GPU)
  // in-launch sampling / weights (FusedCtl::inline_perturb / inline_weights)
  int inl, inw, nu;
  unsigned* cu_state;         // this CU's state word when the chain was taken at the census (partners read it), else NULL
  bool w0;                    // this wave evaluates the costs (wave kCwFused)
  const double* state_in;     // the command's state in the kernel-argument segment (inl)
  const double* U_old;        // control sequence BEFORE the shift (inl)
  const double* u_init;       // in the kernel-argument segment
  const double* pert_g;       // perturbed / bounded noise, published by the encoder tiles of this launch (or an earlier launch)
  const double* noise_g;
  static_assert(NLC_MAX_NU == 2, "two named slots per value below");
  double pc0, pc1, nc0, nc1, pn0, pn1, nn0, nn1;  // (cost wave) perturbed / noise of step t (c) and of step t + 1 (n)
  double pp0, pp1, np0, np1;                      // ... and of step t - 1 (kCwFused != 0: the cost is evaluated one step late)

  __device__ __forceinline__ const double* state0(const RolloutArgs& a, int64_t kc, int ep) const {
    return inl ? state_in : a.state0 + (a.state_per_sample ? kc : (int64_t)ep) * a.net.d;
  }
  __device__ __forceinline__ double pert(const RolloutArgs&, int64_t, int, int j) const { return j == 0 ? pc0 : pc1; }
  __device__ __forceinline__ double noise(const RolloutArgs&, int64_t, int, int i) const { return i == 0 ? nc0 : nc1; }
  __device__ __forceinline__ double pert_prev(const RolloutArgs&, int64_t, int, int j) const { return j == 0 ? pp0 : pp1; }
  __device__ __forceinline__ double noise_prev(const RolloutArgs&, int64_t, int, int i) const { return i == 0 ? np0 : np1; }
  __device__ __forceinline__ double U(const RolloutArgs& a, int uoff, int t, int j) const {
    return inl ? mppi_shifted_U(U_old, u_init, 0, T, nu, t, j) : a.U[uoff + t * a.nu + j];
  }
  __device__ __forceinline__ void store_cost(const RolloutArgs& a, int64_t k, double v) const { a.cost_total[k] = v; }
  __device__ __forceinline__ void load(int t, int64_t kc, double* a0, double* a1) const {
    const v4u v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (int)(((int64_t)t * K + kc) * 16), 0, /*sc1*/ 16);
    *a0 = __builtin_bit_cast(double, ((unsigned long long)v.y << 32) | v.x);
    *a1 = __builtin_bit_cast(double, ((unsigned long long)v.w << 32) | v.z);
  }
  // (wave 0) the sampled action and its bounded noise of step t: L1-bypassing loads, like the latents
  __device__ __forceinline__ void load_pn(int t, int64_t kc, double* p0, double* p1, double* n0, double* n1) const {
    if (!w0) return;
    const int64_t at = (kc * T + t) * nu;
    *p0 = MemSc1::ld(pert_g + at);
    *n0 = MemSc1::ld(noise_g + at);
    if (nu > 1) {
      *p1 = MemSc1::ld(pert_g + at + 1);
      *n1 = MemSc1::ld(noise_g + at + 1);
    }
  }
  // Polling wave: wait until the flag of step t_first is set.  The FIRST look is one 64-lane gather over the flags of
  // the next steps (usually the encoders are ahead and this is the only load for many steps); while waiting only ONE
  // word is polled, with growing sleeps -- 128 workgroups gathering 40 lines each every microsecond was measurable as
  // lost L2 bandwidth for the encoder waves' weight streams.
  __device__ __forceinline__ void wait(int t_first, int t_end, int lane) {
    unsigned naps = 0;
    for (unsigned spins = 0;; ++spins) {
      const int tt = t_first + ((spins == 0 || naps == 0) ? lane : 0);
      unsigned f = 0;
      if (tt < t_end && (naps == 0 || lane == 0)) f = __hip_atomic_load(flags + (int64_t)tt * ntk + tile, NLC_RLX_AGENT);
      const unsigned long long ready = __ballot(f != 0);
      const int cnt = (~ready == 0ull) ? 64 : __builtin_ctzll(~ready);
      if (cnt >= 1) {
        ready_upto = t_first + cnt;
        return;
      }
      if (spins > spin_limit) {
        __hip_atomic_store(sync + kFusedTimeout, 1u + (unsigned)t_first, NLC_RLX_AGENT);
        __hip_atomic_store(timeout_host, 1u + (unsigned)t_first, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        // the give-up word has reached the L2 before this workgroup can count its tile done (the barrier after the poll, then
        // wave 0's ticket): the workgroup that folds the shard's partials reads it and marks them invalid (fused_weight_rank)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        ready_upto = t_end;  // stop polling: the command is lost, the grid must still drain
        return;
      }
      naps = naps < 4 ? naps + 1 : 4;
      if (naps == 1) __builtin_amdgcn_s_sleep(16);
      else if (naps == 2) __builtin_amdgcn_s_sleep(32);
      else __builtin_amdgcn_s_sleep(64);  // ~2 us
    }
  }
  __device__ __forceinline__ void begin(int t0, int wv, int lane, int64_t kc) {
    if (wv == kPollWave) {
      if (t0 >= ready_upto) wait(t0, T, lane);
      stamp_max(sync + kFusedTimeRollBeginFirst, true);
      stamp_max(sync + kFusedTimeRollBeginLast, false);
    }
    __syncthreads();
    load(t0, kc, &cur0, &cur1);
    load_pn(t0, kc, &pc0, &pc1, &nc0, &nc1);
  }
  __device__ __forceinline__ void after_barrier1(int t, int t_end, int wv, int lane) {
    if (wv == kPollWave && t + 1 < t_end && t + 1 >= ready_upto) wait(t + 1, t_end, lane);
  }
  __device__ __forceinline__ void after_barrier2(int t, int t_end, int64_t kc) {
    // progress of this chain for its CU's sleeping partners (one relaxed store per horizon step, every lane the same word)
    if (w0 && cu_state != nullptr) __hip_atomic_store(cu_state, kChainWalking + (unsigned)(t + 1), NLC_RLX_AGENT);
    if (NLC_FUSED_TRACE)  // trace build: when this chain entered the last third of step t
      __hip_atomic_store(sync + kFusedFlags + (int64_t)(T + 1 + t) * ntk + tile, (unsigned)__builtin_amdgcn_s_memrealtime() | 1u,
                         NLC_RLX_AGENT);
    if (t + 1 < t_end) {
      load(t + 1, kc, &nxt0, &nxt1);
      load_pn(t + 1, kc, &pn0, &pn1, &nn0, &nn1);
    }
  }
  __device__ __forceinline__ void advance() {
    cur0 = nxt0;
    cur1 = nxt1;
    pp0 = pc0;
    pp1 = pc1;
    np0 = nc0;
    np1 = nc1;
    pc0 = pn0;
    pc1 = pn1;
    nc0 = nn0;
    nc1 = nn1;
  }
};

// the kernel-argument segment as ordinary (global) memory: per-lane indexed reads of the small arrays that ride in it
__device__ __forceinline__ const FusedArgs* args_in_memory() {
  return (const FusedArgs*)(const void*)__builtin_amdgcn_kernarg_segment_ptr();
}

// Window source of the encoder role when no perturb kernel ran (FusedCtl::inline_perturb): wave 0 samples and bounds the
// B actions of each of the tile's 16 windows -- lane group q takes history entries q, q + 4, ... -- stages the raw GRU
// inputs in LDS (xs[(j * 16 + c) * NLC_MAX_NU + dim]) for the four waves and publishes the entry that belongs to the
// tile's OWN step (the newest one, j = B - 1) as perturbed / noise / actions [k, t].
struct XInline {
  bool on;
  double* xs;
  __device__ __forceinline__ void prepare(const GruArgs& a, int lane, int wv, int64_t kk, int tt, bool valid) {
    if (!on || wv != 0) return;
    const FusedArgs* fa = args_in_memory();
    const PerturbArgs& p = fa->p;
    const int q = lane >> 4, c = lane & 15;
    const int64_t ke = p.k_offset + kk;  // single planner: episode 0
    const bool null_action = p.sample_null_action && (ke == p.K_global - 1);
    for (int j = q; j < a.B; j += 4) {
      const int i = tt + j;  // history index: [action_buffer[1:] ; u_scale * perturbed]
      double v0 = 0.0, v1 = 0.0;
      if (i < a.B - 1) {
        v0 = fa->p.abuf_in[(1 + i) * a.nact];
        if (a.nact > 1) v1 = fa->p.abuf_in[(1 + i) * a.nact + 1];
      } else {
        const int tp = i - (a.B - 1);
        double eps[NLC_MAX_NU];
        mppi_draw(ke, tp, p.seed, p.counter, p.nu, fa->p.mu, fa->p.chol, eps);
#pragma unroll
        for (int dim = 0; dim < NLC_MAX_NU; ++dim) {
          if (dim < a.nact) {
            const double U = mppi_shifted_U(p.U_old, fa->p.u_init, 0, p.T, p.nu, tp, dim);
            const double V = mppi_bound(U, eps[dim], null_action, p.u_scale, p.has_bounds, fa->p.u_min[dim], fa->p.u_max[dim]);
            if (dim == 0) v0 = a.u_scale * V;
            else v1 = a.u_scale * V;
            if (j == a.B - 1 && valid) {
              const int64_t at = (kk * p.T + tp) * p.nu + dim;
              MemSc1::st(p.perturbed + at, V);
              MemSc1::st(p.noise + at, V - U);                                               // :328
              if (p.actions != nullptr) p.actions[at] = (p.u_scale * V) / p.u_scale;          // :255,340
            }
          }
        }
      }
      xs[(j * 16 + c) * NLC_MAX_NU] = v0;
      xs[(j * 16 + c) * NLC_MAX_NU + 1] = v1;
    }
  }
  __device__ __forceinline__ double raw(const GruArgs& a, int64_t wc, int64_t kk, int tt, int j_win, int q, int c,
                                        int ab_off) const {
    if (!on) return XDirect().raw(a, wc, kk, tt, j_win, q, c, ab_off);
    if (q < a.nact) return xs[(j_win * 16 + c) * NLC_MAX_NU + q];
    return (double)(a.B - 1 - j_win);
  }
};

// Wave-level atomics in UNIFORM control flow: every lane executes the atomic, lane 0 adds 1 and the others add 0 (the
// compiler's atomic optimiser folds the 64 same-address adds into one).  Not `if (lane == 0) atomic...` inside the
// persistent loop: a lane-divergent branch next to the loop's back edge lets ROCm 7.2 peel lanes 1..63 into a loop of
// their own that re-reads the ticket lane 0 has not drawn yet and never ends (seen in the ISA and as a hung launch).
__device__ __forceinline__ unsigned wave_ticket(unsigned* ctr, int lane) {
  const unsigned old = __hip_atomic_fetch_add(ctr, lane == 0 ? 1u : 0u, NLC_RLX_AGENT);
  return __builtin_amdgcn_readfirstlane(old);  // lane 0 added first: it holds the pre-add value
}
__device__ __forceinline__ void wave_add_one(unsigned* ctr, int lane) {
  __hip_atomic_fetch_add(ctr, lane == 0 ? 1u : 0u, NLC_RLX_AGENT);
}

// Each role reads the kernel arguments through its OWN laundered copy of the kernel-argument segment pointer (the one
// by-value argument sits at offset 0 of that segment).  Read the ordinary way, the scalar loads of every field either
// role touches are hoisted to the top of the kernel and kept live across both roles: 1500 SGPR spills into VGPR lanes
// (v_readlane / v_writelane + s_nop in the encoder's GEMM loops) and an encoder role 1.5x slower than the stand-alone
// gru_encode_kernel.  Behind the asm statement the loads stay inside the role.
// Tried and dropped: non-inlined role functions reading the segment pointer, and an address-space cast of the by-value
// argument's address (both read address 0 on the MI355X); a copy of the block in global memory read through the
// constant cache (correct, but every scalar load of it crawls: 13 ms per launch instead of 0.8).
typedef const __attribute__((address_space(4))) FusedArgs* fused_args_cptr;
__device__ __forceinline__ fused_args_cptr role_args() {
  fused_args_cptr p = (fused_args_cptr)__builtin_amdgcn_kernarg_segment_ptr();
  asm volatile("" : "+s"(p));
  return p;
}

// The weight folds run once per rollout tile / once per launch: real calls, so that their registers (sixteen loads in
// flight, ocml's exp) are allocated apart from the roles' -- inlined at both rollout call sites they cost the kernel 600
// spilled VGPRs and 30 us per launch (measured).
static __device__ __attribute__((noinline)) void fused_weight_tile(const WeightArgs w, int tile, int lane, double cost, bool valid) {
  weight_tile<MemSc1>(w, 0, (int64_t)tile, lane, cost, valid);
}
// A shard whose launch gave up somewhere (bounded waits) must not be merged: its partial row is marked with eta = -1 (a sum of
// exponentials is never negative), which travels through the shard all-gather, so that merge_kernel on EVERY rank sees it,
// leaves U alone and reports it -- every rank then re-runs the command on the two-launch body (nlc_mppi_finish).
static __device__ __attribute__((noinline)) void fused_weight_rank(const WeightArgs w, double* lds, const unsigned* gave_up) {
  weight_rank<MemSc1>(w, 0, lds);
  if (threadIdx.x == 0 && __hip_atomic_load(gave_up, NLC_RLX_AGENT) != 0u) w.partials[1] = kPartialInvalidEta;  // (thread 0 stored eta)
}

template <int HT, int NT3>
__device__ __forceinline__ void fused_rollout(int tile, double* smem, unsigned* cu_state = nullptr) {
  constexpr int KS = HT * 4;
  const FusedArgs& a = *(const FusedArgs*)role_args();
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  PaHandoff src;
  src.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.r.pa, 0, 0x7fffffff, 0x00020000);
  src.K = a.r.K;
  src.flags = a.ctl.sync + kFusedFlags;
  src.ntk = a.ctl.ntk;
  src.tile = tile;
  src.T = a.r.T;
  src.sync = a.ctl.sync;
  src.timeout_host = a.ctl.timeout_host;
  src.ready_upto = 0;
  src.spin_limit = a.ctl.spin_limit;
  src.inl = a.ctl.inline_perturb;
  src.inw = a.ctl.inline_weights;
  src.nu = a.r.nu;
  src.cu_state = cu_state;
  src.w0 = wv == kCwFused;
  src.state_in = args_in_memory()->p.state_in;
  src.u_init = args_in_memory()->p.u_init;
  src.U_old = a.p.U_old;
  src.pert_g = a.r.perturbed;
  src.noise_g = a.r.noise;
  src.pc0 = src.pc1 = src.nc0 = src.nc1 = src.pn0 = src.pn1 = src.nn0 = src.nn1 = 0.0;
  src.pp0 = src.pp1 = src.np0 = src.np1 = 0.0;
  // the sequential chain is the command's critical path: its waves win the issue arbitration on their SIMDs
  if (NLC_FUSED_TRACE && wv == 0) wave_add_one(a.ctl.sync + kFusedStatRollStart, lane);
  __builtin_amdgcn_s_setprio(3);
  const double cost = rollout_split_tile<HT, NT3, PaHandoff, false, kCwFused, kSplitPrefetchFused>(a.r, (int64_t)tile, src, smem, smem + KS * 64,
                                                                                                     smem + 2 * KS * 64);
  __builtin_amdgcn_s_setprio(0);
  if (a.ctl.inline_weights) {
    // this tile's costs are final: fold its 16 samples (weight_tile), drain the write-through partial, count the tile
    // done (ONE lane adds); the workgroup whose add comes last folds all tile partials into the shard's partials
    int* s_last = reinterpret_cast<int*>(smem);  // (H1 region: dead since the last step's second barrier)
    if (wv == kCwFused) {  // (the wave that holds the tile's costs)
      const bool valid = (int64_t)tile * 16 + (lane & 15) < a.r.K;
      fused_weight_tile(a.w, tile, lane, cost, valid);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      s_last[0] = (int)wave_ticket(a.ctl.sync + kFusedCostDone, lane);
    }
    __syncthreads();
    const bool last = __builtin_amdgcn_readfirstlane(s_last[0]) == a.ctl.ntk - 1;
    __syncthreads();
    if (last) fused_weight_rank(a.w, smem + 8, a.ctl.sync + kFusedTimeout);
    __syncthreads();
  }
  if (wv == 0) {
    if (NLC_FUSED_TRACE) wave_add_one(a.ctl.sync + kFusedStatRollDone, lane);
    stamp_max(a.ctl.sync + kFusedTimeRollEndFirst, true);
    stamp_max(a.ctl.sync + kFusedTimeRollEndLast, false);
  }
}

// Encoder role: the WORKGROUP draws encoder tiles (ticket order = horizon-major: all tiles of step t before step t+1)
// until the ticket is spent, or until it has encoded max_tiles of them (a chain's workgroup before it starts walking),
// and encodes each one cooperatively -- gru_encode_tile_coop: one gate chunk per wavefront, a quarter of the latency of
// a wave-sized tile at the same throughput (at four workgroups per CU), so that work moves between the roles in units of
// ~40 us instead of 150-290.  Wave 0 draws the ticket, applies the head and publishes the tile.
// yield_cu >= 0 (another workgroup of a chain's CU): after yield_after tiles it stops drawing while that chain is
// still running (see the kernel for why).  Never sleeps while it holds a tile: a chain elsewhere may be waiting for it.
template <int G>
__device__ __forceinline__ void fused_encode(double* smem, int max_tiles, int yield_cu, int yield_after) {
  constexpr int KSG = G / 4;
  const FusedArgs& a = *(const FusedArgs*)role_args();
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, c = lane & 15;
  unsigned* sync = a.ctl.sync;
  unsigned* s_ticket = reinterpret_cast<unsigned*>(smem + 4 * KSG * 64);  // behind the four hidden-state images
  XInline xsrc{a.ctl.inline_perturb != 0, smem + 4 * KSG * 64 + 8};      // (+ B x 16 x NLC_MAX_NU staged inputs)
  for (int done = 0; done < max_tiles; ++done) {
    if (wv == 0) {
      if (yield_cu >= 0 && done >= yield_after) {
        // A chain's CU partner (see the kernel).  While the chain walks (state >= kChainWalking + its horizon step) the
        // partner sleeps -- an encoder wave beside a chain doubles the chain's step time -- UNLESS the workgroups on
        // chain-free CUs could not finish the remaining encoder tiles before the chain is through anyway:
        //   remaining tiles * (tile time / chain step time) > remaining chain steps * pool workgroups   -> help.
        // (adaptive = the default schedule: both sides of the inequality shrink as the launch proceeds, so the two roles end
        // together whatever their actual speeds; static schedules -- sleep after a fixed tile count -- for the A/B tools.)
        // Bounded (~50 ms): nothing depends on this wait for correctness.
        for (unsigned spins = 0; spins < (1u << 14); ++spins) {
          const unsigned st = __builtin_amdgcn_readfirstlane(__hip_atomic_load(sync + kFusedCuState + yield_cu, NLC_RLX_AGENT));
          if (st < kChainWalking) break;  // no chain here (any more)
          if (a.ctl.adaptive_q8 > 0) {
            const unsigned drawn = __builtin_amdgcn_readfirstlane(__hip_atomic_load(sync + kFusedEncTicket, NLC_RLX_AGENT));
            const long long rem_tiles = (long long)a.ctl.n_enc - (long long)drawn;
            const long long rem_steps = (long long)a.r.T - (long long)(st - kChainWalking);
            if (rem_tiles <= 0) break;  // nothing left to draw: fall through to the ticket (it ends the role)
            if (rem_tiles * a.ctl.adaptive_q8 > rem_steps * a.ctl.pool_wgs * 256) break;  // the pool alone is too slow: help
          }
          __builtin_amdgcn_s_sleep(127);
        }
      }
      *s_ticket = wave_ticket(sync + kFusedEncTicket, lane);  // (every lane stores the same word)
    }
    __syncthreads();  // also: wave 0 has finished the previous tile's head before anybody zeroes the images again
    const unsigned i = __builtin_amdgcn_readfirstlane(*s_ticket);
    if (i >= (unsigned)a.ctl.n_enc) break;
    const unsigned tr = i / (unsigned)a.ctl.ntk;
    const int t = (int)tr;
    const int j = (int)(i - tr * (unsigned)a.ctl.ntk);
    const int64_t k = (int64_t)j * 16 + c;
    const bool valid = k < a.r.K;
    const int64_t kk = valid ? k : a.r.K - 1;
    const double o = gru_encode_tile_coop<G>(a.g, lane, wv, 0, kk, t, smem, xsrc, valid);
    if (wv == 0 && i != (unsigned)a.ctl.test_drop_tile) {
      if (valid && q < 2) {
        unsigned long long* dst = (unsigned long long*)(a.r.pa + ((int64_t)t * a.r.K + k) * 2 + q);
        __hip_atomic_store(dst, __builtin_bit_cast(unsigned long long, o), NLC_RLX_AGENT);  // global_store_dwordx2 sc1
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the write-through stores have left this wave
      if (NLC_FUSED_TRACE) {
        // trace build: the flag word carries the tile's completion time (any non-zero value publishes the tile)
        __hip_atomic_fetch_or(sync + kFusedFlags + i, (unsigned)__builtin_amdgcn_s_memrealtime() | 1u, NLC_RLX_AGENT);
      } else {
        wave_add_one(sync + kFusedFlags + i, lane);
      }
      if (NLC_FUSED_TRACE) wave_add_one(sync + kFusedStatEncDone, lane);
      stamp_max(sync + kFusedTimeEncLast, false);
    }
  }
}

// BPC = workgroups per CU the instance is compiled for.  4: 128 VGPRs, the rollout role spills 110 of them (its sphere
// map wants ~200) but four cooperative encoder tiles per CU hide each other's latencies -- the better trade when most CUs
// walk a chain (K > 2048 on 256 CUs).  3: 168 VGPRs, 34 spills: chains 5 % faster per horizon step; better up to one
// chain on half the CUs (K = 1024: 0.527 vs 0.571 ms, K = 2048: 0.674 vs 0.709 ms; profiles/r3_fused_small_shard.md).
template <int HT, int NT3, int G, int BPC>
__global__ __launch_bounds__(256, BPC) void nl_plan_fused_kernel(const FusedArgs av) {
  const FusedCtl& a = av.ctl;  // role assignment; the roles read av through role_args()
  constexpr int KSG = G / 4;  // GRU k-steps
  constexpr int KS = HT * 4;  // representation-MLP k-steps
  constexpr int kGruDoubles = 4 * KSG * 64 + 8 + kFusedMaxInlineB * 16 * NLC_MAX_NU, kRollDoubles = 2 * KS * 64 + 8 * 64;
  __shared__ double smem[kGruDoubles > kRollDoubles ? kGruDoubles : kRollDoubles];
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  unsigned* sync = a.sync;
  int* s_tile = reinterpret_cast<int*>(smem);  // broadcast slot (inside wave 0's GRU region / the rollout's H1)

  // ---- census (wave 0, wave-uniform control flow: see wave_ticket): the FIRST workgroup to arrive on a CU may take a
  // rollout tile, so rollout workgroups sit on distinct CUs (two on one CU walk their chains at 19 us per horizon step
  // instead of 10.7: measured, profiles/r2_fused_small_shard.md)
  const unsigned hwid = __builtin_amdgcn_s_getreg((31 << 11) | 4);  // HW_REG_HW_ID: CU_ID [11:8], SH_ID [12], SE_ID [15:13]
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);   // HW_REG_XCC_ID [3:0]
  const unsigned cu = ((xcc & 7u) << 8) | ((hwid >> 8) & 0xffu);
  if (a.inline_perturb && blockIdx.x == 0 && wv == 0) {
    // U <- roll(U, -1); U[-1] = u_init (:199-200): everybody in this launch shifts on the fly, the merge kernel reads this
    const PerturbArgs& p = ((const FusedArgs*)role_args())->p;
    const FusedArgs* fa = args_in_memory();
    for (int i = lane; i < p.T * p.nu; i += 64)
      p.U_new[i] = mppi_shifted_U(p.U_old, fa->p.u_init, 0, p.T, p.nu, i / p.nu, i % p.nu);
  }
  if (wv == 0) {
    int tile = -1;
    if (NLC_FUSED_TRACE) wave_add_one(sync + kFusedStatEntered, lane);
    stamp_max(sync + kFusedTimeEntry, true);
    const unsigned nth = wave_ticket(sync + kFusedCuOcc + cu, lane);
    if (nth == 0) {  // wave-uniform
      const unsigned tk = wave_ticket(sync + kFusedCensusTicket, lane);
      if (tk < (unsigned)a.roll_cap && tk < (unsigned)a.ntk) {
        // ownership of a rollout tile is exclusive: the first add on its owner word (a drain workgroup may get there
        // first when the encoder ticket is dry from the start, tiny K T)
        if (wave_ticket(sync + kFusedFlags + (int64_t)av.r.T * a.ntk + tk, lane) == 0) tile = (int)tk;
      }
      if (tile >= 0) __hip_atomic_store(sync + kFusedCuState + cu, kChainWalking, NLC_RLX_AGENT);  // this CU walks a chain
    }
    s_tile[0] = tile;  // every lane of wave 0 stores the same words
    s_tile[1] = (int)nth;
  }
  __syncthreads();
  int tile = __builtin_amdgcn_readfirstlane(s_tile[0]);
  const int nth = __builtin_amdgcn_readfirstlane(s_tile[1]);
  __syncthreads();

  // ---- schedule (trace build, K = 2048, profiles/r2_fused_small_shard.md): the chains are the launch's critical path --
  // a chain walks a horizon step in 10.7 us alone and in ~21 us beside an encoder wave (an FP64 MFMA holds the SIMD's
  // vector issue for 64 clocks and cannot be pre-empted, whatever the wave priorities), while the chip produces a step's
  // latents in ~17 us.  So a chain starts at once and walks contended while the encoder ticket is young; once n_yield
  // tiles have been drawn the OTHER workgroup of its CU stops drawing and sleeps until the chain is done, the chain
  // finishes at full speed on banked latents, and the CUs without a chain encode the rest.  (Measured and lost: sleeping
  // from the start -- the half chip left produces a step per 19.7 us and starves the chains; chains that start late; the
  // chain's workgroup encoding its own first steps.)
  if (tile >= 0) {
    if (a.chain_first_tiles > 0) {
      fused_encode<G>(smem, a.chain_first_tiles, -1, 0);
      __syncthreads();  // the GRU images in LDS are dead
    }
    fused_rollout<HT, NT3>(tile, smem, sync + kFusedCuState + cu);
    __hip_atomic_store(sync + kFusedCuState + cu, 2u, NLC_RLX_AGENT);  // (all waves, same word) wakes this CU's sleeper
  }

  // ---- encoder role: one tile (horizon step t, samples 16 j .. 16 j + 15) per workgroup and ticket
  __syncthreads();  // (a rollout may just have finished in this LDS)
  fused_encode<G>(smem, 0x7fffffff, (nth != 0 && a.partner_tiles >= 0) ? (int)cu : -1, a.partner_tiles);

  // ---- drain: rollout tiles that have no owner yet (K/16 > roll_cap, or fewer CUs than the host assumed)
  const int first_drain = a.roll_cap < a.ntk ? a.roll_cap : a.ntk;
  for (;;) {
    __syncthreads();
    if (wv == 0) {
      // drain ticket k names tile (roll_cap + k) mod ntk: the tiles the census never offered first, then the offered
      // ones (normally all owned by then: one add each to find out)
      int drawn = -2;  // -2: ticket spent, stop
      const unsigned tk = wave_ticket(sync + kFusedRollTicket, lane);
      if (tk < (unsigned)a.ntk) {
        int cand = first_drain + (int)tk;
        cand = cand >= a.ntk ? cand - a.ntk : cand;
        drawn = wave_ticket(sync + kFusedFlags + (int64_t)av.r.T * a.ntk + cand, lane) == 0 ? cand : -1;  // -1: owned, draw again
      }
      s_tile[0] = drawn;
    }
    __syncthreads();
    tile = __builtin_amdgcn_readfirstlane(s_tile[0]);
    __syncthreads();
    if (tile == -2) break;
    if (tile >= 0) fused_rollout<HT, NT3>(tile, smem);
  }

  if (NLC_FUSED_TRACE && wv == 0) wave_add_one(sync + kFusedStatExited, lane);
}

}  // namespace nlc

// One translation unit per hidden width (kernels_fused.hip: h = 128, the harness's hidden_units; kernels_fused_h64.hip: the
// class default w_nl.py:72; kernels_fused_h256.hip), so the instances compile in parallel.  BPC_LO / BPC_HI: the two
// workgroups-per-CU instances of the width (equal: one instance).
#define NLC_FUSED_DEFINE_LAUNCHERS(SUFFIX, HT_, G_, BPC_LO, BPC_HI)                                                     \
  hipError_t fused_max_resident_blocks_##SUFFIX(int bpc_built, int* blocks_per_cu) {                                   \
    const void* f = bpc_built == BPC_LO ? (const void*)nl_plan_fused_kernel<HT_, 11, G_, BPC_LO>                        \
                                        : (const void*)nl_plan_fused_kernel<HT_, 11, G_, BPC_HI>;                       \
    return hipOccupancyMaxActiveBlocksPerMultiprocessor(blocks_per_cu, f, 256, 0);                                      \
  }                                                                                                                     \
  hipError_t launch_nl_plan_fused_##SUFFIX(const FusedArgs& a, unsigned grid, int bpc_built, hipStream_t s) {          \
    if (bpc_built != BPC_LO && bpc_built != BPC_HI) return hipErrorInvalidValue;                                        \
    switch (a.r.net.nt3) {                                                                                              \
      NLC_FUSED_CASE(7, HT_, G_, BPC_LO, BPC_HI)                                                                        \
      NLC_FUSED_CASE(9, HT_, G_, BPC_LO, BPC_HI)                                                                        \
      NLC_FUSED_CASE(11, HT_, G_, BPC_LO, BPC_HI)                                                                       \
      NLC_FUSED_CASE(13, HT_, G_, BPC_LO, BPC_HI)                                                                       \
      NLC_FUSED_CASE(17, HT_, G_, BPC_LO, BPC_HI)                                                                       \
      NLC_FUSED_CASE(21, HT_, G_, BPC_LO, BPC_HI)                                                                       \
      default:                                                                                                          \
        return hipErrorInvalidValue;                                                                                    \
    }                                                                                                                   \
    return hipGetLastError();                                                                                           \
  }
#define NLC_FUSED_CASE(N, HT_, G_, BPC_LO, BPC_HI)                                                                      \
  case N:                                                                                                               \
    if (bpc_built == BPC_LO) {                                                                                          \
      hipLaunchKernelGGL((nl_plan_fused_kernel<HT_, N, G_, BPC_LO>), dim3(grid), dim3(256), 0, s, a);                   \
    } else {                                                                                                            \
      hipLaunchKernelGGL((nl_plan_fused_kernel<HT_, N, G_, BPC_HI>), dim3(grid), dim3(256), 0, s, a);                   \
    }                                                                                                                   \
    break;

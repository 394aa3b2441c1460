// Device code shared by the rollout kernels (kernels_nl.hip) and the fused planner kernel (kernels_fused.hip):
// one evaluation of the representation MLP + sphere->complex + Fourier ILT (nl_eval), and the latency-split T-step
// rollout of one 16-sample tile by the four waves of a workgroup (rollout_split_tile).  Reference call sites and the
// dataflow are described at the top of kernels_nl.hip.
#pragma once
#include <type_traits>

#include "nlc_device.h"
#include "nlc_envcost.h"
#include "nlc_kernels.h"

// hidden-layer activation of the representation MLP.  The short form of nlc_math.h (tanh_pair_fast: 20 % fewer FP64
// instructions, 1.5e-13 absolute) was measured on one box against the few-ulp tanh_pair_d (round 3, tools/bench_ab.sh with
// -DNLC_HIDDEN_TANH_FAST=1): nl_rollout_kernel 1.234 -> 1.223 ms at K = 16384 (-0.25 % of the command), but the fused
// small-shard body 0.673 -> 0.687 ms at K = 2048 under every static schedule (+2 %): the chains are latency-bound, not
// instruction-bound.  Not shipped.
#ifndef NLC_HIDDEN_TANH_FAST
#define NLC_HIDDEN_TANH_FAST 0
#endif
#if NLC_HIDDEN_TANH_FAST
#define NLC_HIDDEN_TANH_PAIR m::tanh_pair_fast
#else
#define NLC_HIDDEN_TANH_PAIR m::tanh_pair_d
#endif

// -DNLC_PHASE_CLOCKS=1 (tools only: tools/rollout_phase_clocks.py, a separate build of the library): shader-clock stamps at the
// phase boundaries of one model evaluation, summed per wave in SGPRs and added to nlc_phase_clk[] when the wave retires.  The
// product build compiles none of it.
#ifndef NLC_PHASE_CLOCKS
#define NLC_PHASE_CLOCKS 0
#endif

namespace nlc {

struct PhaseClk {
  static constexpr int kN = 10;
  enum { kL1 = 0, kTanh1, kL2, kTanh2, kL3a, kEpiA, kL3b, kEpiB, kTail, kOther };
  uint64_t last;
  uint64_t acc[kN];
  __device__ __forceinline__ void start() {
    if (!NLC_PHASE_CLOCKS) return;
#pragma unroll
    for (int i = 0; i < kN; ++i) acc[i] = 0;
    last = __builtin_amdgcn_s_memtime();
  }
  __device__ __forceinline__ void mark(int i) {
    if (!NLC_PHASE_CLOCKS) return;
    __builtin_amdgcn_sched_barrier(0);
    const uint64_t now = __builtin_amdgcn_s_memtime();
    acc[i] += now - last;
    last = now;
    __builtin_amdgcn_sched_barrier(0);
  }
};
#if NLC_PHASE_CLOCKS
extern __device__ unsigned long long nlc_phase_clk[16];  // kernels_nl.hip
#endif

// The same for the LATENCY-SPLIT bodies (round 5: rollout_split_tile -- stand-alone and as the chain of the fused one-launch body --
// and repfunc_split_mlp of the staged de Hoog planner): four waves share a 16-sample tile and meet at two / three workgroup barriers
// per model evaluation, so next to the arithmetic phases the stamps separate what a wave WAITS for -- the barriers, the hand-over of
// latents from another workgroup -- from what it computes.  Every translation unit that instantiates such a kernel owns a copy of the
// sums (static: no relocatable device code) and exports a reader (NLC_DEFINE_SPLIT_CLK_READER; tools/split_phase_clocks.py).
struct SplitClk {
  static constexpr int kN = 12;
  enum { kHead = 0, kL1, kBar1, kL2, kTanh2, kBar2, kL3, kEpi, kBar3, kTail, kWait, kOther };
  uint64_t last;
  uint64_t acc[kN];
  __device__ __forceinline__ void start() {
    if (!NLC_PHASE_CLOCKS) return;
#pragma unroll
    for (int i = 0; i < kN; ++i) acc[i] = 0;
    last = __builtin_amdgcn_s_memtime();
  }
  __device__ __forceinline__ void mark(int i) {
    if (!NLC_PHASE_CLOCKS) return;
    __builtin_amdgcn_sched_barrier(0);
    const uint64_t now = __builtin_amdgcn_s_memtime();
    acc[i] += now - last;
    last = now;
    __builtin_amdgcn_sched_barrier(0);
  }
};
#if NLC_PHASE_CLOCKS
// [wave index 0..3][phase 0..11] sums, then [48 + wave] = waves counted, [52] = model evaluations (tile-steps) of wave 0
static __device__ unsigned long long nlc_split_clk[64];
// (`sampled`: only one workgroup in 64 adds its sums -- a launch of one-evaluation workgroups would otherwise spend its time in
// 50 k same-line atomics, and every phase would measure the L2 queueing behind them)
__device__ __forceinline__ void split_clk_flush(const SplitClk& clk, int wv, int lane, unsigned long long evals, bool sampled = false) {
  if (lane != 0 || (sampled && (blockIdx.x & 63) != 0)) return;
  for (int i = 0; i < SplitClk::kN; ++i) atomicAdd(&nlc_split_clk[(wv & 3) * SplitClk::kN + i], (unsigned long long)clk.acc[i]);
  atomicAdd(&nlc_split_clk[48 + (wv & 3)], 1ull);
  if ((wv & 3) == 0) atomicAdd(&nlc_split_clk[52], evals);
}
#define NLC_DEFINE_SPLIT_CLK_READER(name)                                                                                  \
  extern "C" int name(unsigned long long* out64) {                                                                         \
    if (hipDeviceSynchronize() != hipSuccess) return -1;                                                                   \
    if (hipMemcpyFromSymbol(out64, HIP_SYMBOL(nlc::nlc_split_clk), 64 * sizeof(unsigned long long)) != hipSuccess) return -1; \
    const unsigned long long zero[64] = {0};                                                                               \
    return hipMemcpyToSymbol(HIP_SYMBOL(nlc::nlc_split_clk), zero, sizeof(zero)) == hipSuccess ? 0 : -1;                   \
  }
#else
#define NLC_DEFINE_SPLIT_CLK_READER(name)
#endif

// ------------------------------------------------------------------ one model evaluation
// p0/p1: layer-1 latent B fragments (index 4s+q).  Returns acc_x: ILT sums, rows = dims (reg r -> dim q+4r).
// |F| times the component the Fourier phase i^k keeps, as (num * trig, den): even-k groups need cos(theta), odd-k
// groups sin(theta) = -cos(theta + pi/2).  Short forms of nlc_math.h (one reduction by pi with the quarter turn in
// the reduction, Chebyshev cosines): 34 FP64 instructions instead of 56 for the fdlibm-kernel pair.
__device__ __forceinline__ double sphere_term(double theta, double phi, bool odd) {
  const m::IltTrigK K = m::ilt_trig_k();
  double num, den;
  m::tan_parts_short(K, phi / 2.0 + kPi / 4.0, &num, &den);
  const double c = m::cos_plus_mpio2(K, theta, odd ? 0.5 : 0.0, odd ? 1.0 : 0.0);
  const double trig = odd ? -c : c;
  return (num * trig) * m::rcp_refined(den);
}

// The module's two angles from the raw layer-3 outputs (w_nl.py:59-62) and the tangent's argument, with every product and sum a
// separate operation (a saturated phi must be EXACTLY pi/2 to reproduce the reference's clamped |F|, and the de Hoog path's
// kernels -- staged launches, persistent chain -- must agree to the bit whatever the compiler would fuse around them).
__device__ __forceinline__ double sphere_theta(double y) {
#pragma clang fp contract(off)
  return m::tanh_d(y) * kPi;
}
__device__ __forceinline__ double sphere_phi(double y) {
#pragma clang fp contract(off)
  return m::tanh_d(y) * kPi / 2.0 - kPi / 2.0 + kPi / 2.0;
}
__device__ __forceinline__ double sphere_tan_arg(double phi) {
#pragma clang fp contract(off)
  return phi / 2.0 + kPi / 4.0;
}

// LIN instances (fixed Talbot / Stehfest): both components of F = R e^{i theta}
__device__ __forceinline__ void sphere_terms_lin(double theta, double phi, double* rc, double* rs) {
  const m::IltTrigK K = m::ilt_trig_k();
  double num, den, sn, cs;
  m::tan_parts_short(K, phi / 2.0 + kPi / 4.0, &num, &den);
  m::sincos_plus_mpio2(K, theta, 0.0, 0.0, &sn, &cs);
  const double rad = num * m::rcp_refined(den);
  *rc = rad * cs;
  *rs = rad * sn;
}

// GENERAL_T: sphere coordinates of the per-sample query points enter layer 1 through W1s.
// FOUT != nullptr-mode (WRITE_F): instead of the Fourier sum, F_k = |F| e^{i theta} of every Laplace term is
// written to (N, d, S) arrays for the de Hoog kernel (nonlinear in F, cannot be an MFMA).
struct FOut {
  double* fre;
  double* fim;
  const int* slot;  // slot index (8 per layer-3 tile) -> c*S + k, -1 = padding
  int64_t row;      // sample row; < 0: do not store
  int dS;
  int angles;       // 1: store (theta, phi) instead of (Re F, Im F)
  int64_t n_rows;   // > 0: slot-major (8*nt3, n_rows) arrays -- slot 4g+q of sample `row` at (4g+q)*n_rows + row, so one
                    // store instruction writes four 128-B runs (16 consecutive samples per lane group q)
};
// NLC_EVAL_PIPELINE (A/B switch, tools/bench_ab.sh): where the loads that are NOT a GEMM's own k-step fragments are issued --
// bias tiles, each GEMM's first fragments, the ILT coefficient tiles, a horizon loop's next layer 1.
//   0  at the head of the phase that consumes them (rounds 1-3)
//   1  inside the k loop of the GEMM BEFORE that phase, a few per k-step, scheduled into the MFMA shadows together with the
//      k-step's own fragments (gemm_acc_head's `extra`).  One wave per SIMD has nobody to hide behind, and a vector load costs
//      its wave ~17 issue clocks wherever no MFMA is running (tools/rollout_phase_clocks.py: moving the loads in front of the
//      activation phase made that phase longer by exactly their issue time).
// Arithmetic, MFMA order per tile and ILT sum order are the same in both: same bits.
// NLC_L3_SINGLE_PASS: 1 = layer 3 as one GEMM over its NT3 output tiles, 0 = two halves (round 3).
#ifndef NLC_EVAL_PIPELINE
#define NLC_EVAL_PIPELINE 1
#endif
#ifndef NLC_L3_SINGLE_PASS
#define NLC_L3_SINGLE_PASS 0
#endif
// Where the SMALL tables of the network are read from (bias vectors, layer-1 fragments, ILT coefficient tiles: 23 KB at the
// headline shape) -- every access returns a pointer the optimiser cannot see through, so nothing is hoisted out of a horizon loop.
//   NlTabsGlobal  the packed arrays in HBM / L2 (NlNetArgs)
//   NlTabsLds     a copy the workgroup made in LDS at kernel start (nl_rollout_kernel): an LDS read costs its wave a fraction
//                 of a vector load's issue time and latency, and these loads sit at the HEAD of every phase, where one wave per
//                 SIMD has nothing to hide them behind
struct NlTabsGlobal {
  __device__ __forceinline__ const double* b1(const NlNetArgs& n) const { return (const double*)opaque(n.b1); }
  __device__ __forceinline__ const double* b2(const NlNetArgs& n) const { return (const double*)opaque(n.b2); }
  __device__ __forceinline__ const double* b3p(const NlNetArgs& n) const { return (const double*)opaque(n.b3p); }
  __device__ __forceinline__ gptr w1(const NlNetArgs& n) const { return opaque(n.W1p); }
  __device__ __forceinline__ gptr cp(const NlNetArgs& n) const { return opaque(n.Cp); }
  __device__ __forceinline__ const double* w2(const NlNetArgs& n) const { return n.W2p; }  // (laundered by the GEMM)
};
// W2: layer 2's whole fragment-packed matrix as well (h^2 doubles: 128 KB at hidden_units 128 -- with the small tables that is
// 154 KB of the CU's 160 KB, one workgroup per CU, which the kernel's one wave per SIMD implies anyway)
template <int HT, int NT3, bool W2 = false>
struct NlTabsLds {
  static constexpr int kB1 = 0, kB2 = 16 * HT, kB3 = 32 * HT, kW1 = kB3 + 16 * NT3, kCp = kW1 + 2 * HT * 64;
  static constexpr int kW2 = kCp + 2 * NT3 * 64;
  static constexpr int kDoubles = kW2 + (W2 ? HT * HT * 4 * 64 : 0);
  lptr base;
  __device__ __forceinline__ auto w2(const NlNetArgs& n) const {
    if constexpr (W2)
      return (lptr)(base + kW2);
    else
      return n.W2p;
  }
  __device__ __forceinline__ lptr b1(const NlNetArgs&) const { return opaque_lds(base + kB1); }
  __device__ __forceinline__ lptr b2(const NlNetArgs&) const { return opaque_lds(base + kB2); }
  __device__ __forceinline__ lptr b3p(const NlNetArgs&) const { return opaque_lds(base + kB3); }
  __device__ __forceinline__ lptr w1(const NlNetArgs&) const { return opaque_lds(base + kW1); }
  __device__ __forceinline__ lptr cp(const NlNetArgs&) const { return opaque_lds(base + kCp); }
  // all threads of the workgroup, before its first barrier
  __device__ __forceinline__ static void fill(double* sm, const NlNetArgs& n, int tid, int nthreads) {
    for (int i = tid; i < 16 * HT; i += nthreads) {
      sm[kB1 + i] = n.b1[i];
      sm[kB2 + i] = n.b2[i];
    }
    for (int i = tid; i < 16 * NT3; i += nthreads) sm[kB3 + i] = n.b3p[i];
    for (int i = tid; i < 2 * HT * 64; i += nthreads) sm[kW1 + i] = n.W1p[i];
    for (int i = tid; i < 2 * NT3 * 64; i += nthreads) sm[kCp + i] = n.Cp[i];
    if constexpr (W2) {
      for (int i = tid; i < HT * HT * 4 * 64; i += nthreads) sm[kW2 + i] = n.W2p[i];
    }
  }
};

// Layer 1 of one model evaluation: bias tiles and both k-steps' weight fragments -- what a horizon loop loads one step ahead.
template <int HT>
struct NlL1Pre {
  v4d bias[HT];
  double w1[2 * HT];
  static constexpr int kItems = 6 * HT;
  // item I < 4 HT: register I & 3 of bias tile I >> 2; then the 2 HT fragments.  b1q = b1 + q, both pointers laundered.
  template <int I, class BP, class WP>
  __device__ __forceinline__ void item(BP b1q, WP w, int lane) {
    if constexpr (I < 4 * HT)
      bias[I >> 2][I & 3] = b1q[16 * (I >> 2) + 4 * (I & 3)];
    else if constexpr (I < kItems)
      w1[I - 4 * HT] = w[(I - 4 * HT) * 64 + lane];
  }
  template <class TABS = NlTabsGlobal>
  __device__ __forceinline__ void load(const NlNetArgs& n, int lane, int q, const TABS& tabs = TABS{}) {
    const auto b1q = tabs.b1(n) + q;
    const auto w = tabs.w1(n);
    static_for<kItems>([&](auto ic) { item<decltype(ic)::value>(b1q, w, lane); });
  }
};

struct NlNoPre {};  // a single evaluation: layer 1 is loaded where it is used

// sph_row (GENERAL_T only): when non-NULL, this lane's sample has EXPLICIT sphere inputs [theta_s | phi_s] (2S doubles)
// -- LaplaceRepresentationFunc.forward on an arbitrary input row (w_nl.py:55-63) -- instead of the ones of s_k(tn).
// PRE = NlL1Pre<HT>: `pre` holds this evaluation's layer 1 and receives the next one's; NlNoPre: neither.  (A type, not a
// nullable pointer: a null test on the caller's register struct would pin it to scratch memory.)
template <int HT, int NT3, bool GENERAL_T, bool WRITE_F, bool LIN, class PRE, class TABS = NlTabsGlobal>
__device__ __forceinline__ v4d nl_eval_impl(const NlNetArgs& n, int lane, int q, double p0, double p1, double tn, const FOut* fo,
                                            const double* sph_row, PhaseClk* pc, PRE& pre, const TABS& tabs = TABS{}) {
  constexpr int KS = HT * 4;  // h / 4
  // (wider output layers, hidden_units 256: the registers are needed elsewhere)
  constexpr bool PIPE = NLC_EVAL_PIPELINE != 0 && NT3 <= 17 && HT <= 8;
  constexpr bool HAS_PRE = !std::is_same<PRE, NlNoPre>::value;
  auto phase = [&](int i) {
    if (NLC_PHASE_CLOCKS && pc != nullptr) pc->mark(i);
  };
  v4d h1[HT];
  double w1[2 * HT];
  if constexpr (HAS_PRE) {
#pragma unroll
    for (int j = 0; j < HT; ++j) h1[j] = pre.bias[j];
#pragma unroll
    for (int i = 0; i < 2 * HT; ++i) w1[i] = pre.w1[i];
  } else {
    NlL1Pre<HT> now;
    now.load(n, lane, q, tabs);
#pragma unroll
    for (int j = 0; j < HT; ++j) h1[j] = now.bias[j];
#pragma unroll
    for (int i = 0; i < 2 * HT; ++i) w1[i] = now.w1[i];
  }
  if constexpr (GENERAL_T) {
    // s_k = gamma + i pi k / T,  T = scale*t,  gamma = alpha - ln(tol)/(scale*T); theta_s = atan2(Im, Re),
    // phi_s = asin((|s|^2-1)/(|s|^2+1)); input order [theta_s(0..S-1) | phi_s(0..S-1)]
    const double Tt = n.scale * tn;
    const double gamma = n.alpha - n.log_tol / (n.scale * Tt);
    const int kss = (2 * n.S + 3) / 4;
    gptr p = opaque(n.W1s);
    for (int ks = 0; ks < kss; ++ks) {
      const int i = 4 * ks + q;
      double b = 0.0;
      if (sph_row != nullptr) {
        if (i < 2 * n.S) b = sph_row[i];
      } else if (i < 2 * n.S) {
        const int k = (i < n.S) ? i : i - n.S;
        const double im = kPi * (double)k / Tt;
        if (i < n.S) {
          b = atan2(im, gamma);
        } else {
          const double a2 = gamma * gamma + im * im;
          b = asin((a2 - 1.0) / (a2 + 1.0));
        }
      }
#pragma unroll
      for (int m = 0; m < HT; ++m) h1[m] = mfma(p[m * 64 + lane], b, h1[m]);
      p = opaque(p + HT * 64);
    }
  }

  // ---- layer 1 (fragments already in registers); in its shadow: layer 2's bias tiles and first fragments
  v4d h2[HT];
  double a2[HT];
  {
    const auto b2q = tabs.b2(n) + q;
    const auto w2 = opaque(tabs.w2(n));
    auto item2 = [&](auto ic) {
      constexpr int i = decltype(ic)::value;
      if constexpr (i < 4 * HT)
        h2[i >> 2][i & 3] = b2q[16 * (i >> 2) + 4 * (i & 3)];
      else if constexpr (i < 5 * HT)
        a2[i - 4 * HT] = w2[(i - 4 * HT) * 64 + lane];
    };
    constexpr int IPM = 3;  // 5 HT items over 2 HT MFMAs
    static_for<2 * HT>([&](auto mc) {
      constexpr int km = decltype(mc)::value, ks = km / HT, m = km % HT;
      h1[m] = mfma(w1[ks * HT + m], ks == 0 ? p0 : p1, h1[m]);
      if (PIPE) {
        static_for<IPM>([&](auto e) { item2(std::integral_constant<int, km * IPM + decltype(e)::value>{}); });
#if NLC_GEMM_INTERLEAVE
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x020, IPM, 0);
#endif
      }
    });
    if (PIPE) __builtin_amdgcn_sched_barrier(0);
    phase(PhaseClk::kL1);
    // Activations are applied as one batch of 32 independent tanh chains per layer.  (On gfx950 an FP64 MFMA
    // holds the SIMD's VALU issue for its whole 64 cycles -- tools/ubench_f64.hip -- so interleaving the
    // activation with the next layer's MFMAs buys nothing, while batching keeps the FP64 VALU latency hidden.)
#pragma unroll
    for (int j = 0; j < HT; ++j)
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        double ta, tb;
        NLC_HIDDEN_TANH_PAIR(h1[j][r], h1[j][r + 1], &ta, &tb);
        h1[j][r] = ta;
        h1[j][r + 1] = tb;
      }
    phase(PhaseClk::kTanh1);
    if (!PIPE) static_for<5 * HT>(item2);
  }

  // ---- layer 2; in its shadow: the bias tiles, first fragments and ILT coefficient tiles of layer 3's first pass.
  // Layer 3 runs in two passes over its output tiles (round 3): half the accumulators and prefetch registers live at a time.
  constexpr int NA = NLC_L3_SINGLE_PASS ? NT3 : (NT3 + 1) / 2, NB = NT3 - NA, NB1 = NB > 0 ? NB : 1;
  // coefficient tiles fetched ahead per output tile (not pipelined: read where they are used, as LIN's second table always is)
  constexpr int NCP = (WRITE_F || !PIPE) ? 0 : 2;
  v4d oa[NA], ob[NB1];
  double a3a[NA], a3b[NB1], cpa[2 * NA], cpb[2 * NB1];
  const auto b3q = tabs.b3p(n) + q;
  gptr w3 = opaque(n.W3p);
  const auto cp = tabs.cp(n);
  auto item3a = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if constexpr (i < 4 * NA)
      oa[i >> 2][i & 3] = b3q[16 * (i >> 2) + 4 * (i & 3)];
    else if constexpr (i < 5 * NA)
      a3a[i - 4 * NA] = w3[(i - 4 * NA) * 64 + lane];
    else if constexpr (i < (5 + NCP) * NA)
      cpa[i - 5 * NA] = cp[(i - 5 * NA) * 64 + lane];
  };
  auto item3b = [&](auto ic) {
    constexpr int i = decltype(ic)::value;
    if constexpr (NB > 0) {
      if constexpr (i < 4 * NB)
        ob[i >> 2][i & 3] = b3q[16 * (NA + (i >> 2)) + 4 * (i & 3)];
      else if constexpr (i < 5 * NB)
        a3b[i - 4 * NB] = w3[(NA + i - 4 * NB) * 64 + lane];
      else if constexpr (i < (5 + NCP) * NB)
        cpb[i - 5 * NB] = cp[(2 * NA + i - 5 * NB) * 64 + lane];
    }
  };
  constexpr int N3A = (5 + NCP) * NA, N3B = (5 + NCP) * NB;
  constexpr int E2 = PIPE ? (N3A + KS - 1) / KS : 0;
  gemm_acc_head<HT, HT, KS, E2>(h2, a2, tabs.w2(n), 0, lane, [&](int ks) { return h1[ks >> 2][ks & 3]; }, [&](auto kc) {
    static_for<E2>([&](auto e) { item3a(std::integral_constant<int, decltype(kc)::value * E2 + decltype(e)::value>{}); });
  });
  phase(PhaseClk::kL2);
#pragma unroll
  for (int j = 0; j < HT; ++j)
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      double ta, tb;
      NLC_HIDDEN_TANH_PAIR(h2[j][r], h2[j][r + 1], &ta, &tb);
      h2[j][r] = ta;
      h2[j][r + 1] = tb;
    }
  phase(PhaseClk::kTanh2);
  if (!PIPE) static_for<N3A>(item3a);

  v4d ax[1];
  ax[0] = splat(0.0);
  auto epilogue = [&](auto& o, auto& cpv, auto J0, auto NJ) {
#pragma unroll
    for (int jj = 0; jj < decltype(NJ)::value; ++jj) {
      const int j = decltype(J0)::value + jj;
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const int g = 2 * j + r;
        // (not paired: a saturated phi must be EXACTLY pi/2 to reproduce the reference's clamped |F|)
        const double theta = sphere_theta(o[jj][r]);    // w_nl.py:59
        const double phi = sphere_phi(o[jj][r + 2]);    // w_nl.py:60-62
        // |F| (cos theta | sin theta) = num/den * cos(theta - [odd] pi/2); the division is folded into the product
        double num, den;
        m::tan_parts_0_halfpi(sphere_tan_arg(phi), &num, &den);
        if constexpr (WRITE_F) {
          const int idx = fo->slot[4 * g + q];
          if (fo->row >= 0 && idx >= 0 && fo->angles) {
            fo->fre[fo->row * fo->dS + idx] = theta;  // the module's own outputs (theta, phi), w_nl.py:59-63
            fo->fim[fo->row * fo->dS + idx] = phi;
          } else if (fo->row >= 0 && idx >= 0) {
            double sn, cs;
            m::sincos_bounded(theta, &sn, &cs);
            const double rad = num * m::rcp_refined(den);
            const int64_t at = fo->n_rows > 0 ? (int64_t)(4 * g + q) * fo->n_rows + fo->row : fo->row * fo->dS + idx;
            fo->fre[at] = rad * cs;
            fo->fim[at] = rad * sn;
          }
        } else {
          (void)num;
          (void)den;
          if constexpr (LIN) {
            double rc, rs;
            sphere_terms_lin(theta, phi, &rc, &rs);
            ax[0] = mfma(NCP ? cpv[2 * jj + r] : cp[g * 64 + lane], rc, ax[0]);
            ax[0] = mfma(opaque(n.Cp2)[g * 64 + lane], rs, ax[0]);
          } else {
            ax[0] = mfma(NCP ? cpv[2 * jj + r] : cp[g * 64 + lane], sphere_term(theta, phi, g >= n.n_even_groups), ax[0]);
          }
        }
      }
    }
  };
  // a horizon loop's next evaluation: its layer-1 tiles ride in the shadow of this one's last GEMM
  const auto b1q = tabs.b1(n) + q;
  const auto w1n = tabs.w1(n);
  constexpr int NL1 = NlL1Pre<HT>::kItems;
  constexpr int EL1 = PIPE ? (NL1 + KS - 1) / KS : 0;
  auto next_l1 = [&](auto kc) {
    if constexpr (HAS_PRE)
      static_for<EL1>([&](auto e) { pre.template item<decltype(kc)::value * EL1 + decltype(e)::value>(b1q, w1n, lane); });
  };
  if constexpr (NB > 0) {
    // ---- layer 3, first pass; in its shadow: the second pass's tiles
    constexpr int E3A = PIPE ? (N3B + KS - 1) / KS : 0;
    gemm_acc_head<NA, NT3, KS, E3A>(oa, a3a, n.W3p, 0, lane, [&](int ks) { return h2[ks >> 2][ks & 3]; }, [&](auto kc) {
      static_for<E3A>([&](auto e) { item3b(std::integral_constant<int, decltype(kc)::value * E3A + decltype(e)::value>{}); });
    });
    phase(PhaseClk::kL3a);
    epilogue(oa, cpa, std::integral_constant<int, 0>{}, std::integral_constant<int, NA>{});
    phase(PhaseClk::kEpiA);
    if (!PIPE) static_for<N3B>(item3b);
    // ---- layer 3, second pass; in its shadow: the next evaluation's layer 1
    gemm_acc_head<NB1, NT3, KS, EL1>(ob, a3b, n.W3p, NA, lane, [&](int ks) { return h2[ks >> 2][ks & 3]; }, next_l1);
    phase(PhaseClk::kL3b);
    epilogue(ob, cpb, std::integral_constant<int, NA>{}, std::integral_constant<int, NB>{});
    phase(PhaseClk::kEpiB);
  } else {
    gemm_acc_head<NA, NT3, KS, EL1>(oa, a3a, n.W3p, 0, lane, [&](int ks) { return h2[ks >> 2][ks & 3]; }, next_l1);
    phase(PhaseClk::kL3a);
    epilogue(oa, cpa, std::integral_constant<int, 0>{}, std::integral_constant<int, NA>{});
    phase(PhaseClk::kEpiA);
  }
  if constexpr (!PIPE && HAS_PRE) pre.load(n, lane, q, tabs);
  return ax[0];
}
template <int HT, int NT3, bool GENERAL_T, bool WRITE_F = false, bool LIN = false>
__device__ __forceinline__ v4d nl_eval(const NlNetArgs& n, int lane, int q, double p0, double p1, double tn,
                                       const FOut* fo = nullptr, const double* sph_row = nullptr, PhaseClk* pc = nullptr) {
  NlNoPre none;
  return nl_eval_impl<HT, NT3, GENERAL_T, WRITE_F, LIN>(n, lane, q, p0, p1, tn, fo, sph_row, pc, none);
}

// ---- the k loops of the latency-split bodies (round 5).  Four waves share a tile, so a k-step is only TW = 2 (layer 2) or
// NTW = 2 .. 6 (layer 3) MFMAs per wave -- 128 .. 192 clocks at hidden_units 128 -- while a weight fragment takes ~250 clocks from
// the L2.  split_gemm<.., D> keeps D k-steps of fragments in flight (a ring of D x NT registers, every index a compile-time
// constant of the unrolled loop); D = 1 is the loop of rounds 1-4 (fragments of step ks + 1 requested at step ks).  Same MFMAs in
// the same order: bit-identical results for every D.  MEASURED (same box, K = 2048 / 8192 / cfg5; tools/split_ab.py): the
// stand-alone split rollout and the de Hoog planner's representation launch do not move for D = 2 .. 6 (the compiler already
// hoists ~5 k-steps of loads to the loop head; D = 4 costs cfg5 4 % through the launch's registers); the fused body's chains,
// whose fragment loads compete with the encoder role's weight streams, gain 1.3 % at D = 4 (675.6 -> 667.1 us per launch):
// D = 4 there, 1 elsewhere.
#ifdef NLC_SPLIT_PREFETCH
constexpr int kSplitPrefetch = NLC_SPLIT_PREFETCH, kSplitPrefetchFused = NLC_SPLIT_PREFETCH;
#else
constexpr int kSplitPrefetch = 1, kSplitPrefetchFused = 4;
#endif
// acc[i] += sum_ks A_frag(ks, i) * Hb[ks]: `base` = fragment (ks = 0, tile 0) of this wave's tiles, `step` doubles between k-steps,
// off(i) = offset in doubles of the wave's i-th output tile (compile-time i), Hb = the (KS, 64) activation image in LDS.
template <int NT, int KS, int D, class OFF>
__device__ __forceinline__ void split_gemm(v4d (&acc)[NT], gptr base, const int step, OFF off, const double* __restrict__ Hb,
                                           const int lane) {
  if constexpr (D <= 1) {
    gptr p = base;
    double a_cur[NT], a_nxt[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) a_cur[i] = p[off(i) + lane];
    double b_cur = Hb[lane], b_nxt = 0.0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) {
        p = opaque(p + step);
#pragma unroll
        for (int i = 0; i < NT; ++i) a_nxt[i] = p[off(i) + lane];
        b_nxt = Hb[(ks + 1) * 64 + lane];
      }
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = mfma(a_cur[i], b_cur, acc[i]);
#pragma unroll
      for (int i = 0; i < NT; ++i) a_cur[i] = a_nxt[i];
      b_cur = b_nxt;
    }
  } else {
    constexpr int DD = D < KS ? D : KS;
    double ring[DD][NT];
    gptr p = base;
#pragma unroll
    for (int st = 0; st < DD; ++st) {
      if (st > 0) p = opaque(p + step);
#pragma unroll
      for (int i = 0; i < NT; ++i) ring[st][i] = p[off(i) + lane];
    }
    double b_cur = Hb[lane], b_nxt = 0.0;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) b_nxt = Hb[(ks + 1) * 64 + lane];
#pragma unroll
      for (int i = 0; i < NT; ++i) acc[i] = mfma(ring[ks % DD][i], b_cur, acc[i]);
      if (ks + DD < KS) {  // the slot just consumed takes the fragments of step ks + D
        p = opaque(p + step);
#pragma unroll
        for (int i = 0; i < NT; ++i) ring[ks % DD][i] = p[off(i) + lane];
      }
      b_cur = b_nxt;
    }
  }
}

// Which of the four waves evaluates the running cost / perturbation cost of a sample (rollout_split_tile's CW).  0 (rounds 1-4):
// wave 0, right after the state update -- 1.6 k clocks per step during which the other three waves already wait at the next step's
// first barrier, while the wave with one layer-3 tile less (wave 3: tiles j = wave + 4 i, and no instantiated NT3 is a multiple of
// four) idles 2.3 k clocks at the THIRD barrier of every step (tools/split_phase_clocks.py, profiles/r5_split_phase_clocks.md).
// 3: that wave evaluates the cost of step t - 1 inside its idle time of step t (the state entering step t IS the state after step
// t - 1; the sampled action / noise of t - 1 are read again or kept one step longer), and the cost of the last step after the loop.
// Same operations on the same operands in the same order: bit-identical.  Measured (K = 2048 / 8192, T = 40, same box):
// nl_rollout_split_kernel 425.8 -> 402.9 us / 714 -> 694 us; the fused one-launch body, whose chains are paced by the encoder
// role beside them, does not gain (675.6 -> 683.9 us: the extra live values cost registers) and keeps wave 0.
constexpr int kSplitCostWaveStandalone = 3, kSplitCostWaveFused = 0;
// (tools: -DNLC_SPLIT_COST_WAVE=n / -DNLC_SPLIT_PREFETCH=n override both bodies' settings for A/B builds, tools/build_split_variants.sh)
#ifdef NLC_SPLIT_COST_WAVE
constexpr int kCwStandalone = NLC_SPLIT_COST_WAVE, kCwFused = NLC_SPLIT_COST_WAVE;
#else
constexpr int kCwStandalone = kSplitCostWaveStandalone, kCwFused = kSplitCostWaveFused;
#endif

// ------------------------------------------------------------------ latency-split rollout of one 16-sample tile
// nl_rollout_kernel gives every wavefront a whole 16-sample tile, which fills the chip only when K/16 >= 1024
// wavefronts.  When the population is sharded over several GPUs (K/G = 2048 at 8 GPUs) most SIMDs would idle
// while 40 strictly sequential horizon steps run at single-wave speed.  Here ONE workgroup (4 waves = the 4
// SIMDs of a CU) owns the 16-sample tile and splits every layer's OUTPUT tiles over its waves; the full
// activation vector is exchanged through LDS ([k-step][lane] images, conflict-free ds_read_b64/ds_write_b64),
// three barriers per horizon step.  Per-step latency drops ~4x; total MFMA work is unchanged.
// (The k loops here are left to the compiler's schedule: forcing the MFMA / load interleave that pays in the wave-per-tile
// kernel -- gemm_kstep_order -- cost this one 13 %, 0.428 -> 0.482 ms at K = 2048, and the fused body 7 %; handing wave 0's
// layer-1 tiles to the other three waves, so that its state-and-cost tail overlaps their layer 1, gained 1.9 % here at K <= 4096
// but lost 1.5 % at 8192 and 3 % in the fused body, where the SIMD time of the recomputed surplus tile is an encoder wave's; round 4.)
//
// Where the GRU latents of a step come from is a policy:
//   PaDirect   the (K, T, 2) tensor an earlier launch wrote (nl_rollout_split_kernel)
//   PaHandoff  (kernels_fused.hip) tiles published by encoder workgroups of the SAME launch, behind per-tile flags
// Policy interface (all calls are made by all four waves, in this order per horizon step t):
//   state0(a, kc, ep)            where the rollout's start state is read from
//   begin(t0, wv, lane, kc)      before the loop; leaves the latents of step t0 in cur0/cur1
//   after_barrier1(t, ...)       between the first and second barrier of step t   (poll for step t+1)
//   after_barrier2(t, ...)       after the second barrier of step t               (issue the loads of step t+1)
//   pert / noise / U (wave 0)    the sampled action, its bounded noise and the nominal control of step t
//   advance()                    end of step t: the latents of step t+1 become cur0/cur1
//   store_cost(a, k, v)          (wave 0, lane group 0) the sample's total cost
struct PaDirect {
  const double* pa;
  int T;
  double cur0, cur1, nxt0, nxt1;
  __device__ __forceinline__ void load(int t, int64_t kc, double* a0, double* a1) const {
    const double* p = pa + (kc * T + t) * 2;
    *a0 = p[0];
    *a1 = p[1];
  }
  __device__ __forceinline__ const double* state0(const RolloutArgs& a, int64_t kc, int ep) const {
    return a.state0 + (a.state_per_sample ? kc : (int64_t)ep) * a.net.d;
  }
  __device__ __forceinline__ double pert(const RolloutArgs& a, int64_t kc, int t, int j) const {
    return a.perturbed[(kc * a.T + t) * a.nu + j];
  }
  __device__ __forceinline__ double noise(const RolloutArgs& a, int64_t kc, int t, int i) const {
    return a.noise[(kc * a.T + t) * a.nu + i];
  }
  // (deferred cost, CW != 0: the values of an EARLIER step t are simply read at their index)
  __device__ __forceinline__ double pert_prev(const RolloutArgs& a, int64_t kc, int t, int j) const { return pert(a, kc, t, j); }
  __device__ __forceinline__ double noise_prev(const RolloutArgs& a, int64_t kc, int t, int i) const { return noise(a, kc, t, i); }
  __device__ __forceinline__ double U(const RolloutArgs& a, int uoff, int t, int j) const { return a.U[uoff + t * a.nu + j]; }
  __device__ __forceinline__ void store_cost(const RolloutArgs& a, int64_t k, double v) const { a.cost_total[k] = v; }
  __device__ __forceinline__ void begin(int t0, int, int, int64_t kc) { load(t0, kc, &cur0, &cur1); }
  __device__ __forceinline__ void after_barrier1(int, int, int, int) {}
  __device__ __forceinline__ void after_barrier2(int t, int t_end, int64_t kc) {
    if (t + 1 < t_end) load(t + 1, kc, &nxt0, &nxt1);
  }
  __device__ __forceinline__ void advance() {
    cur0 = nxt0;
    cur1 = nxt1;
  }
};

// Returns the sample's total cost (meaningful in wave CW when the launch runs the last horizon chunk).
template <int HT, int NT3, class PA, bool LIN = false, int CW = kCwStandalone, int D = kSplitPrefetch>
__device__ __forceinline__ double rollout_split_tile(const RolloutArgs& a, int64_t tile, PA& src, double* __restrict__ H1,
                                                     double* __restrict__ H2, double* __restrict__ AX) {
  constexpr int KS = HT * 4;           // k-steps over the hidden dimension
  constexpr int TW = HT / 4;           // layer-1/2 output tiles per wave
  constexpr int NTW = (NT3 + 3) / 4;   // layer-3 output tiles per wave (tile j = wave + 4 i)
  const NlNetArgs& n = a.net;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, c = lane & 15;
  const int64_t k = tile * 16 + c;
  const bool valid = k < a.K;
  const int64_t kc = valid ? k : a.K - 1;
  const int d = n.d;
  const int i0 = q, i1 = 4 + q;
  double x0 = 0.0, x1 = 0.0, m0 = 0.0, m1 = 0.0, s0 = 1.0, s1 = 1.0;
  const bool first_chunk = a.t_begin == 0, last_chunk = a.t_end == a.T;
  const int ep = (int)(kc / a.Kep);  // episode of this lane's sample (0 for the single planner)
  const int uoff = ep * a.T * a.nu;
  const double* st = first_chunk ? src.state0(a, kc, ep) : a.xcarry + kc * d;
  if (i0 < d) {
    x0 = st[i0];
    m0 = n.state_mean[i0];
    s0 = n.state_std[i0];
  }
  if (i1 < d) {
    x1 = st[i1];
    m1 = n.state_mean[i1];
    s1 = n.state_std[i1];
  }
  const double Tt = n.scale * a.tn;
  const double gamma = n.alpha - n.log_tol / (n.scale * Tt);
  const double factor = LIN ? 1.0 : exp(gamma * a.tn) / Tt;  // (LIN: 1 / t rides in the coefficient tables)
  // this wave's layer-3 tiles (clamped: a wave with fewer tiles recomputes the last one and drops it)
  int j3[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i) j3[i] = (wv + 4 * i < NT3) ? wv + 4 * i : NT3 - 1;

  double cost = 0.0, pcost = 0.0;
  if (!first_chunk) {
    cost = a.ccarry[kc * 2];
    pcost = a.ccarry[kc * 2 + 1];
  }
  // running cost (mppi_with_model.py:145-171) and perturbation cost (:343-344) of horizon step tc from the state AFTER that step,
  // which the calling wave holds in (x0, x1).  `earlier`: called one step late (CW != 0), the source hands back the
  // sampled action / noise of the step before its current one.
  auto step_cost = [&](int tc, bool earlier) {
    double xs[NLC_MAX_D];
#pragma unroll
    for (int i = 0; i < NLC_MAX_D; ++i) xs[i] = __shfl((i < 4) ? x0 : x1, ((i & 3) << 4) | c, 64);
    double u[NLC_MAX_NU] = {0.0, 0.0};
    double pc = 0.0;
    for (int j = 0; j < a.nu; ++j) u[j] = a.u_scale * (earlier ? src.pert_prev(a, kc, tc, j) : src.pert(a, kc, tc, j));
    for (int j = 0; j < a.nu; ++j) {
      double acj = 0.0;
      for (int i = 0; i < a.nu; ++i) {
        double e = earlier ? src.noise_prev(a, kc, tc, i) : src.noise(a, kc, tc, i);
        if (a.noise_abs_cost) e = fabs(e);
        acj += (a.lambda_ * e) * a.sigma_inv[i * a.nu + j];
      }
      pc += src.U(a, uoff, tc, j) * acj;
    }
    cost += running_cost(a.env, xs, u, a.nu);
    pcost += pc;
  };
  SplitClk clk;
  clk.start();
  src.begin(a.t_begin, wv, lane, kc);
  clk.mark(SplitClk::kWait);
  for (int t = a.t_begin; t < a.t_end; ++t) {
    const double pa0 = src.cur0, pa1 = src.cur1;
    const double p0 = (i0 < d) ? (x0 - m0) / s0 : (i0 == d ? pa0 : (i0 == d + 1 ? pa1 : 0.0));
    const double p1 = (i1 < d) ? (x1 - m1) / s1 : (i1 == d ? pa0 : (i1 == d + 1 ? pa1 : 0.0));
    clk.mark(SplitClk::kHead);
    // ---- layer 1: output tiles TW*wv .. TW*wv+TW-1
    {
      v4d acc[TW];
#pragma unroll
      for (int i = 0; i < TW; ++i) acc[i] = load_bias_tile(n.b1, TW * wv + i, q);
      gptr p = opaque(n.W1p + (size_t)TW * wv * 64);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const double b = ks == 0 ? p0 : p1;
#pragma unroll
        for (int i = 0; i < TW; ++i) acc[i] = mfma(p[(ks * HT + i) * 64 + lane], b, acc[i]);
      }
#pragma unroll
      for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          double ta, tb;
          NLC_HIDDEN_TANH_PAIR(acc[i][r], acc[i][r + 1], &ta, &tb);
          H1[(4 * (TW * wv + i) + r) * 64 + lane] = ta;
          H1[(4 * (TW * wv + i) + r + 1) * 64 + lane] = tb;
        }
    }
    clk.mark(SplitClk::kL1);
    __syncthreads();
    clk.mark(SplitClk::kBar1);
    src.after_barrier1(t, a.t_end, wv, lane);
    clk.mark(SplitClk::kWait);
    // ---- layer 2
    {
      v4d acc[TW];
#pragma unroll
      for (int i = 0; i < TW; ++i) acc[i] = load_bias_tile(n.b2, TW * wv + i, q);
      split_gemm<TW, KS, D>(acc, opaque(n.W2p + (size_t)TW * wv * 64), HT * 64, [](int i) { return i * 64; }, H1, lane);
      clk.mark(SplitClk::kL2);
#pragma unroll
      for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          double ta, tb;
          NLC_HIDDEN_TANH_PAIR(acc[i][r], acc[i][r + 1], &ta, &tb);
          H2[(4 * (TW * wv + i) + r) * 64 + lane] = ta;
          H2[(4 * (TW * wv + i) + r + 1) * 64 + lane] = tb;
        }
    }
    clk.mark(SplitClk::kTanh2);
    __syncthreads();
    clk.mark(SplitClk::kBar2);
    src.after_barrier2(t, a.t_end, kc);
    // ---- layer 3 (own tiles) + sphere->complex + partial ILT sum
    v4d ax = splat(0.0);
    {
      v4d o[NTW];
#pragma unroll
      for (int i = 0; i < NTW; ++i) o[i] = load_bias_tile(n.b3p, j3[i], q);
      split_gemm<NTW, KS, D>(o, opaque(n.W3p), NT3 * 64, [&](int i) { return j3[i] * 64; }, H2, lane);
      clk.mark(SplitClk::kL3);
      gptr cp = opaque(n.Cp);
#pragma unroll
      for (int i = 0; i < NTW; ++i) {
        if (wv + 4 * i < NT3) {  // wave-uniform
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int g = 2 * j3[i] + r;
            const double theta = m::tanh_d(o[i][r]) * kPi;
            const double phi = m::tanh_d(o[i][r + 2]) * kPi / 2.0 - kPi / 2.0 + kPi / 2.0;
            if constexpr (LIN) {
              double rc, rs;
              sphere_terms_lin(theta, phi, &rc, &rs);
              ax = mfma(cp[g * 64 + lane], rc, ax);
              ax = mfma(opaque(n.Cp2)[g * 64 + lane], rs, ax);
            } else {
              ax = mfma(cp[g * 64 + lane], sphere_term(theta, phi, g >= n.n_even_groups), ax);
            }
          }
        }
      }
    }
    AX[(wv * 2 + 0) * 64 + lane] = ax[0];
    AX[(wv * 2 + 1) * 64 + lane] = ax[1];
    clk.mark(SplitClk::kEpi);
    // (x0, x1) still hold the state after step t - 1: its cost, in the time this wave would wait at the barrier below
    if (CW != 0 && wv == CW && t > a.t_begin) step_cost(t - 1, true);
    clk.mark(SplitClk::kTail);
    __syncthreads();
    clk.mark(SplitClk::kBar3);
    // every wave forms the same full sums (fixed order) and keeps its own copy of the state
    const double ax0 = ((AX[0 * 64 + lane] + AX[2 * 64 + lane]) + AX[4 * 64 + lane]) + AX[6 * 64 + lane];
    const double ax1 = ((AX[1 * 64 + lane] + AX[3 * 64 + lane]) + AX[5 * 64 + lane]) + AX[7 * 64 + lane];
    if (i0 < d) x0 = x0 + factor * ax0;
    if (i1 < d) x1 = x1 + factor * ax1;
    if (wv == 0 && valid && a.states != nullptr) {
      double* so = a.states + (k * a.T + t) * d;
      if (i0 < d) so[i0] = x0;
      if (i1 < d) so[i1] = x1;
    }
    if (CW == 0 && wv == 0) step_cost(t, false);
    src.advance();
    clk.mark(SplitClk::kTail);
  }
  if (CW != 0 && wv == CW && a.t_end > a.t_begin) step_cost(a.t_end - 1, true);  // the last step's
#if NLC_PHASE_CLOCKS
  split_clk_flush(clk, wv, lane, (unsigned long long)(a.t_end - a.t_begin));
#endif
  if (wv == CW && valid) {
    if (last_chunk) {
      if (q == 0) src.store_cost(a, k, cost + pcost);
    } else {
      if (i0 < d) a.xcarry[k * d + i0] = x0;
      if (i1 < d) a.xcarry[k * d + i1] = x1;
      if (q == 0) {
        a.ccarry[k * 2] = cost;
        a.ccarry[k * 2 + 1] = pcost;
      }
    }
  }
  return cost + pcost;
}

// ------------------------------------------------------------------ latency-split representation function (de Hoog path)
// One workgroup = one 16-sample tile, as rollout_split_tile: every layer's output tiles are split over the four waves and
// the activations are exchanged through LDS -- but the output is F_k itself (slot-major, for ilt_dehoog_kernel), written
// by the wave that owns the layer-3 tile, so there is no third barrier.  The staged de Hoog planner launches this once per
// horizon step: a wave-sized tile (nl_repfunc_kernel) does the whole MLP of 16 samples serially, one wave per SIMD, and
// is latency-bound for the 75 us the launch lasts; four waves per tile at up to four workgroups per CU overlap.
// Planner form only (constant query time folded into b1, slot-major F, optional tail of the previous step).
//
// The MLP of repfunc_split_tile from the layer-1 B fragments on: p0 / p1 = this lane's latent entries q and 4 + q of sample c
// (normalised state dims, then the two GRU latents), `k` the sample's column in the slot-major F arrays of `n_cols` columns.
// after_l1() runs behind the first barrier (the staged kernel stores the carried state there).  `wv`: the wave's index within
// the FOUR waves that share the tile -- a workgroup of eight waves may run two tiles side by side, each group with LDS images
// of its own and the same barrier sequence (kernels_dehoog_chain.hip).  The weights must be reachable through kernel-argument
// (wave-uniform) pointers: `n` is a reference into the kernel's argument block, never a local copy.
template <int HT, int NT3, class AfterL1>
__device__ __forceinline__ void repfunc_split_mlp(const NlNetArgs& n, const double p0, const double p1, const bool valid,
                                                  const int64_t k, const int64_t n_cols, const int* __restrict__ slot,
                                                  double* __restrict__ fre, double* __restrict__ fim, double* __restrict__ H1,
                                                  double* __restrict__ H2, const int wv, const int lane, AfterL1 after_l1,
                                                  SplitClk& clk) {
  constexpr int KS = HT * 4;
  constexpr int TW = HT / 4;
  constexpr int NTW = (NT3 + 3) / 4;
  const int q = lane >> 4;
  // (the bias tiles are loaded through laundered pointers: inside a persistent horizon loop the compiler would otherwise hoist
  // these loop-invariant loads and keep ~80 VGPRs live across the QD phase of kernels_dehoog_chain.hip)
  int j3[NTW];
#pragma unroll
  for (int i = 0; i < NTW; ++i) j3[i] = (wv + 4 * i < NT3) ? wv + 4 * i : NT3 - 1;
  // ---- layer 1: output tiles TW*wv .. TW*wv+TW-1
  {
    v4d acc[TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) acc[i] = load_bias_tile((const double*)opaque(n.b1), TW * wv + i, q);
    gptr p = opaque(n.W1p + (size_t)TW * wv * 64);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const double b = ks == 0 ? p0 : p1;
#pragma unroll
      for (int i = 0; i < TW; ++i) acc[i] = mfma(p[(ks * HT + i) * 64 + lane], b, acc[i]);
    }
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        double ta, tb;
        NLC_HIDDEN_TANH_PAIR(acc[i][r], acc[i][r + 1], &ta, &tb);
        H1[(4 * (TW * wv + i) + r) * 64 + lane] = ta;
        H1[(4 * (TW * wv + i) + r + 1) * 64 + lane] = tb;
      }
  }
  clk.mark(SplitClk::kL1);
  __syncthreads();
  clk.mark(SplitClk::kBar1);
  after_l1();
  // ---- layer 2
  {
    v4d acc[TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) acc[i] = load_bias_tile((const double*)opaque(n.b2), TW * wv + i, q);
    split_gemm<TW, KS, kSplitPrefetch>(acc, opaque(n.W2p + (size_t)TW * wv * 64), HT * 64, [](int i) { return i * 64; }, H1, lane);
    clk.mark(SplitClk::kL2);
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        double ta, tb;
        NLC_HIDDEN_TANH_PAIR(acc[i][r], acc[i][r + 1], &ta, &tb);
        H2[(4 * (TW * wv + i) + r) * 64 + lane] = ta;
        H2[(4 * (TW * wv + i) + r + 1) * 64 + lane] = tb;
      }
  }
  clk.mark(SplitClk::kTanh2);
  __syncthreads();
  clk.mark(SplitClk::kBar2);
  // ---- layer 3 (own tiles) + sphere -> complex, F_k stored slot-major
  {
    v4d o[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) o[i] = load_bias_tile((const double*)opaque(n.b3p), j3[i], q);
    split_gemm<NTW, KS, kSplitPrefetch>(o, opaque(n.W3p), NT3 * 64, [&](int i) { return j3[i] * 64; }, H2, lane);
    clk.mark(SplitClk::kL3);
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      if (wv + 4 * i < NT3) {  // wave-uniform
#pragma unroll
        for (int r = 0; r < 2; ++r) {
          const int g = 2 * j3[i] + r;
          // (same arithmetic as nl_eval's WRITE_F branch)
          const double theta = sphere_theta(o[i][r]);
          const double phi = sphere_phi(o[i][r + 2]);
          double num, den;
          m::tan_parts_0_halfpi(sphere_tan_arg(phi), &num, &den);
          const int idx = slot[4 * g + q];
          if (valid && idx >= 0) {
            double sn, cs;
            m::sincos_bounded(theta, &sn, &cs);
            const double rad = num * m::rcp_refined(den);
            const int64_t at = (int64_t)(4 * g + q) * n_cols + k;
            fre[at] = rad * cs;
            fim[at] = rad * sn;
          }
        }
      }
    }
  }
  clk.mark(SplitClk::kEpi);
}

// The same MLP for NS 16-sample tiles at once by a workgroup of NW = 2 NS waves (hidden_units 128: HT = 8), the form the
// persistent de Hoog step chain uses (kernels_dehoog_chain.hip; NS = 4: eight waves, NS = 2: four): wave w owns output tiles
// TW w .. TW w + TW - 1 (TW = 8 / NW) of layers 1 and 2 and output tiles w, w + NW, w + 2 NW, ... of layer 3 -- for ALL NS sample
// tiles, so every weight fragment fetched from L2 feeds NS MFMAs instead of one (the 4-wave form streams the whole 466 KB weight
// set once per 16 samples).  Per output tile and sample tile the MFMA sequence over k is the one of repfunc_split_mlp and the
// activations are the same functions: the F values are the same bits.
// p0 / p1 [s]: layer-1 fragments of sample tile s for this lane's column; column of sample (s, c) in the F block: 16 s + c of
// 16 NS.  H1 / H2: [NS][32 * 64] LDS images.  Two workgroup barriers.
template <int NT3, int NS>
__device__ __forceinline__ void repfunc_block_mlp(const NlNetArgs& n, const double (&p0)[NS], const double (&p1)[NS],
                                                  const int rows_here, const int* __restrict__ slot, double* __restrict__ fre,
                                                  double* __restrict__ fim, double* __restrict__ H1, double* __restrict__ H2,
                                                  const int wave, const int lane) {
  constexpr int HT = 8, KS = 32, NW = 2 * NS, TW = HT / NW;
  constexpr int NTW = (NT3 + NW - 1) / NW;  // layer-3 output tiles per wave (tile j = wave + NW i)
  static_assert(NS == 2 || NS == 4, "two or four sample tiles per workgroup");
  const int q = lane >> 4, c = lane & 15;
  // ---- layer 1: output tiles TW wave + i
  {
    v4d acc[TW][NS];
#pragma unroll
    for (int i = 0; i < TW; ++i) {
      const v4d bias = load_bias_tile((const double*)opaque(n.b1), TW * wave + i, q);
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) acc[i][s2] = bias;
    }
    gptr p = opaque(n.W1p + (size_t)TW * wave * 64);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < TW; ++i) {
        const double a = p[(ks * HT + i) * 64 + lane];
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) acc[i][s2] = mfma(a, ks == 0 ? p0[s2] : p1[s2], acc[i][s2]);
      }
    }
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          double ta, tb;
          NLC_HIDDEN_TANH_PAIR(acc[i][s2][r], acc[i][s2][r + 1], &ta, &tb);
          H1[s2 * KS * 64 + (4 * (TW * wave + i) + r) * 64 + lane] = ta;
          H1[s2 * KS * 64 + (4 * (TW * wave + i) + r + 1) * 64 + lane] = tb;
        }
  }
  __syncthreads();
  // ---- layer 2: output tiles TW wave + i
  {
    v4d acc[TW][NS];
#pragma unroll
    for (int i = 0; i < TW; ++i) {
      const v4d bias = load_bias_tile((const double*)opaque(n.b2), TW * wave + i, q);
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) acc[i][s2] = bias;
    }
    gptr p = opaque(n.W2p + (size_t)TW * wave * 64);
    double a_cur[TW], a_nxt[TW];
#pragma unroll
    for (int i = 0; i < TW; ++i) {
      a_cur[i] = p[i * 64 + lane];
      a_nxt[i] = 0.0;
    }
    double b_cur[NS], b_nxt[NS];
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
      b_cur[s2] = H1[s2 * KS * 64 + lane];
      b_nxt[s2] = 0.0;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) {
        p = opaque(p + HT * 64);
#pragma unroll
        for (int i = 0; i < TW; ++i) a_nxt[i] = p[i * 64 + lane];
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) b_nxt[s2] = H1[s2 * KS * 64 + (ks + 1) * 64 + lane];
      }
#pragma unroll
      for (int i = 0; i < TW; ++i)
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) acc[i][s2] = mfma(a_cur[i], b_cur[s2], acc[i][s2]);
#pragma unroll
      for (int i = 0; i < TW; ++i) a_cur[i] = a_nxt[i];
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) b_cur[s2] = b_nxt[s2];
    }
#pragma unroll
    for (int i = 0; i < TW; ++i)
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          double ta, tb;
          NLC_HIDDEN_TANH_PAIR(acc[i][s2][r], acc[i][s2][r + 1], &ta, &tb);
          H2[s2 * KS * 64 + (4 * (TW * wave + i) + r) * 64 + lane] = ta;
          H2[s2 * KS * 64 + (4 * (TW * wave + i) + r + 1) * 64 + lane] = tb;
        }
  }
  __syncthreads();
  // ---- layer 3 (own tiles, all NS sample tiles) + sphere -> complex, F_k stored slot-major
  {
    int j3[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) j3[i] = (wave + NW * i < NT3) ? wave + NW * i : NT3 - 1;
    v4d o[NTW][NS];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      const v4d bias = load_bias_tile((const double*)opaque(n.b3p), j3[i], q);
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) o[i][s2] = bias;
    }
    gptr p = opaque(n.W3p);
    double a_cur[NTW], a_nxt[NTW];
#pragma unroll
    for (int i = 0; i < NTW; ++i) a_cur[i] = p[j3[i] * 64 + lane];
    double b_cur[NS], b_nxt[NS];
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
      b_cur[s2] = H2[s2 * KS * 64 + lane];
      b_nxt[s2] = 0.0;
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if (ks + 1 < KS) {
        p = opaque(p + NT3 * 64);
#pragma unroll
        for (int i = 0; i < NTW; ++i) a_nxt[i] = p[j3[i] * 64 + lane];
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) b_nxt[s2] = H2[s2 * KS * 64 + (ks + 1) * 64 + lane];
      }
#pragma unroll
      for (int i = 0; i < NTW; ++i)
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) o[i][s2] = mfma(a_cur[i], b_cur[s2], o[i][s2]);
#pragma unroll
      for (int i = 0; i < NTW; ++i) a_cur[i] = a_nxt[i];
#pragma unroll
      for (int s2 = 0; s2 < NS; ++s2) b_cur[s2] = b_nxt[s2];
    }
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
      if (wave + NW * i < NT3) {  // wave-uniform
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2)
#pragma unroll
          for (int r = 0; r < 2; ++r) {
            const int g = 2 * j3[i] + r;
            // (same arithmetic as repfunc_split_mlp / nl_eval's WRITE_F branch)
            const double theta = sphere_theta(o[i][s2][r]);
            const double phi = sphere_phi(o[i][s2][r + 2]);
            double num, den;
            m::tan_parts_0_halfpi(sphere_tan_arg(phi), &num, &den);
            const int idx = slot[4 * g + q];
            if (16 * s2 + c < rows_here && idx >= 0) {
              double sn, cs;
              m::sincos_bounded(theta, &sn, &cs);
              const double rad = num * m::rcp_refined(den);
              const int64_t at = (int64_t)(4 * g + q) * (16 * NS) + 16 * s2 + c;
              fre[at] = rad * cs;
              fim[at] = rad * sn;
            }
          }
      }
    }
  }
}

// the staged planner's per-step launch: observation (or the tail of the previous step) -> layer-1 fragments -> the MLP above
template <int HT, int NT3>
__device__ __forceinline__ void repfunc_split_tile(const RepFuncArgs& a, int64_t tile, double* __restrict__ H1,
                                                   double* __restrict__ H2) {
  const NlNetArgs& n = a.net;
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = lane >> 4, c = lane & 15;
  const int64_t k = tile * 16 + c;
  const bool valid = k < a.N;
  const int64_t kc = valid ? k : a.N - 1;
  const int d = n.d;
  const int i0 = q, i1 = 4 + q;
  SplitClk clk;
  clk.start();
  const double* ob = a.obs + (a.obs_per_sample ? kc : kc / a.Kep) * a.obs_stride;
  const double* pa = a.pa + kc * a.pa_stride;
  double x0 = (i0 < d) ? ob[i0] : 0.0, x1 = (i1 < d) ? ob[i1] : 0.0;
  if (a.tail_prev) {
    // tail of the previous horizon step, as in nl_repfunc_kernel; every wave needs the new state, wave 0 stores it
    const StepTailArgs& s = a.tail;
    const int64_t e = kc / s.Kep;
    const double* src = s.first ? s.state0 + (s.state_per_sample ? kc : e) * d : s.x + kc * d;
    if (i0 < d) x0 = src[i0] + s.dx[kc * d + i0];
    if (i1 < d) x1 = src[i1] + s.dx[kc * d + i1];
    if (wv == 0 && valid && s.states != nullptr) {
      if (i0 < d) s.states[(k * s.T + s.t) * d + i0] = x0;
      if (i1 < d) s.states[(k * s.T + s.t) * d + i1] = x1;
    }
    // (the carried state s.x is stored BEHIND the first barrier below: the other three waves read it above, and a
    // store from here could overtake their loads -- seen as wrong states from the third horizon step on once another
    // stream's kernels shared the CUs, round 3)
    // (the previous step's COSTS are evaluated at the END of this function, by wave kTailCostWave: round 5)
  }
  const double p0 = (i0 < d) ? (x0 - n.state_mean[i0]) / n.state_std[i0]
                             : (i0 == d ? pa[0] : (i0 == d + 1 ? pa[1] : 0.0));
  const double p1 = (i1 < d) ? (x1 - n.state_mean[i1]) / n.state_std[i1]
                             : (i1 == d ? pa[0] : (i1 == d + 1 ? pa[1] : 0.0));
  clk.mark(SplitClk::kHead);
  repfunc_split_mlp<HT, NT3>(n, p0, p1, valid, k, a.N, a.slot, a.fre, a.fim, H1, H2, wv, lane, [&]() {
    if (a.tail_prev && wv == 0 && valid) {  // every wave has read the previous state: now it may be replaced
      if (i0 < d) a.tail.x[k * d + i0] = x0;
      if (i1 < d) a.tail.x[k * d + i1] = x1;
    }
  }, clk);
  // Running cost and perturbation cost of the previous horizon step (mppi_with_model.py:145-171, mppi_delay.py:343-344), from the
  // state this workgroup started from.  Rounds 2-4 evaluated them in wave 0 BEFORE layer 1: 10 k clocks during which the other three
  // waves waited at the first barrier -- 4 us of the launch's 56 at cfg5's size, in the wave that also owns the extra layer-3 tile
  // (tools/split_phase_clocks.py, profiles/r5_split_phase_clocks.md).  Nothing in this launch needs them: the wave with the fewest
  // layer-3 tiles (wave 3; no instantiated NT3 is a multiple of four) evaluates them behind its own epilogue.  Same operations.
  constexpr int kTailCostWave = 3;
  if (a.tail_prev && wv == kTailCostWave) {
    const StepTailArgs& s = a.tail;
    const int64_t e = kc / s.Kep;
    double xs[NLC_MAX_D];
#pragma unroll
    for (int i = 0; i < NLC_MAX_D; ++i) xs[i] = __shfl((i < 4) ? x0 : x1, ((i & 3) << 4) | c, 64);
    if (q == 0 && valid) {
      double u[NLC_MAX_NU] = {0.0, 0.0};
      for (int j = 0; j < s.nu; ++j) u[j] = s.u_scale * s.perturbed[(k * s.T + s.t) * s.nu + j];
      const double pc = perturbation_cost_step(s.noise + (k * s.T + s.t) * s.nu, s.U + (e * s.T + s.t) * s.nu, s.sigma_inv,
                                               s.lambda_, s.nu, s.noise_abs_cost);
      s.ccarry[k * 2] = (s.first ? 0.0 : s.ccarry[k * 2]) + running_cost(s.env, xs, u, s.nu);
      s.ccarry[k * 2 + 1] = (s.first ? 0.0 : s.ccarry[k * 2 + 1]) + pc;
    }
  }
  clk.mark(SplitClk::kTail);
#if NLC_PHASE_CLOCKS
  split_clk_flush(clk, wv, lane, 1ull, true);
#endif
}

}  // namespace nlc

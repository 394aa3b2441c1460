// The step chain of the de Hoog planner (BASELINE configs[4]: NeuralLaplaceModel.forward with ilt_algorithm = "dehoog",
// w_nl.py:137-144, inside the T sequential rollout steps of planners/mppi_delay.py:271-296) as ONE persistent launch.
//
// The staged path (abi_planner_nl.hip: rollout_nl_staged) runs 2 T + 1 launches per command -- representation function -> F_k
// slot-major in HBM, de Hoog ILT -> dx, with the state / cost tail folded into the next representation launch -- on two streams.
// Here a workgroup of eight wavefronts OWNS 64 consecutive samples (four 16-sample MFMA tiles) for the whole horizon and walks
// their chain alone: no inter-workgroup dependency exists (samples are independent; the GRU latents come from the hoisted
// encode launch), so the only synchronisation is the workgroup barrier.  Per horizon step:
//   phase A   the representation MLP of the four tiles in ONE pass (repfunc_block_mlp: wave w owns output tile w of the hidden
//             layers and tiles w, w + 8, w + 16 of layer 3 for all four sample tiles, so a weight fragment fetched from L2 feeds
//             four MFMAs; same MFMA sequence per tile and same activations as the staged path's repfunc_split_tile), F_k written
//             slot-major into the workgroup's private (8 nt3) x 64 block;
//   phase B   the QD table: wave p < d owns dim p of the 64 samples (dehoog_row, the staged path's own code; every load one
//             full 512-B line of the block, which this CU wrote a few microseconds ago), dx added to the state in LDS;
//   tail      wave 0, one lane per sample: state store, running cost and perturbation cost of the step (StepTailArgs semantics).
// Same arithmetic, in the same order, as the staged path: bit-identical states, costs and actions (tests).
// F never leaves the chip's caches as far as the kernel can tell: 169 KB per workgroup and step, written and re-read by the
// same CU (the staged path moves the same 43 MB per step through two launches).  It cannot live in LDS: 64 samples x 165 terms
// x 16 B = 169 KB, and a 16-sample tile's QD is 80 rows -- 1.25 wavefronts -- so a tile-local chain would run the FP64-VALU-bound
// half at a third of the lanes.
#include "nlc_dehoog_row.h"
#include "nlc_nl_kernels.h"

namespace nlc {

// NS = sample tiles per workgroup.  NS = 4: eight waves own 64 samples, one workgroup per CU (256 VGPRs x 8 waves), the QD phase is
// one wavefront per dim.  NS = 2 (round 4, second form): four waves own 32 samples and TWO workgroups share a CU -- independent
// chains that drift apart, so one's VALU-bound QD phase runs beside the other's MFMA-bound representation phase; a QD wavefront
// then holds two dims' rows (lanes 0-31 / 32-63) and the idle fourth wave rotates over the SIMDs from step to step.
template <int NT3, int M, int NS>
__global__ __launch_bounds__(128 * NS, NS == 4 ? 1 : 2) void nl_dehoog_chain_kernel(const DehoogChainArgs a) {
  constexpr int HT = 8, KS = HT * 4, S = 2 * M + 1, NW = 2 * NS, SPB = 16 * NS;  // SPB: samples per block
  constexpr int CH = M > 8 ? M + 1 : 9;  // terms fetched at a time (as ilt_dehoog_kernel)
  __shared__ double H1[NS][KS * 64], H2[NS][KS * 64];
  __shared__ double XS[SPB * NLC_MAX_D];  // the block's states, [sample][dim]
  const NlNetArgs& n = a.net;
  const int d = n.d;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int64_t nblk = (a.K + SPB - 1) / SPB;
  const double t = a.tn;
  const double Tt = n.scale * t;
  const double gamma = n.alpha - n.log_tol / (n.scale * Tt);
  const double ang = kPi * (t / Tt);
  const cplx z = {cos(ang), sin(ang)};
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t k0 = blk * SPB;
    const int rows_here = (int)((a.K - k0 < SPB) ? (a.K - k0) : SPB);
    double* fre = a.fre + (size_t)blk * SPB * 8 * NT3;
    double* fim = a.fim + (size_t)blk * SPB * 8 * NT3;
    // start state of every sample of the block
    for (int i = threadIdx.x; i < SPB * NLC_MAX_D; i += 64 * NW) {
      const int sm = i / NLC_MAX_D, dim = i % NLC_MAX_D;
      const int64_t k = k0 + (sm < rows_here ? sm : rows_here - 1);
      XS[i] = dim < d ? a.state0[(a.state_per_sample ? k : 0) * d + dim] : 0.0;
    }
    double cost = 0.0, pcost = 0.0;  // wave 0: lane = sample
    __syncthreads();
    for (int t_h = 0; t_h < a.T; ++t_h) {
      // ---- phase A: representation function of the block's NS tiles in one pass (every weight fragment feeds NS MFMAs)
      if (a.phases & 1) {
        const int q = lane >> 4, c = lane & 15, i0 = q, i1 = 4 + q;
        double p0[NS], p1[NS];
#pragma unroll
        for (int s2 = 0; s2 < NS; ++s2) {
          const int kb = 16 * s2 + c;  // sample within the block
          const int kc = kb < rows_here ? kb : rows_here - 1;
          const double* pa = a.pa + ((size_t)(k0 + kc) * a.T + t_h) * 2;
          const double* xr = XS + kc * NLC_MAX_D;
          const double x0 = (i0 < d) ? xr[i0] : 0.0, x1 = (i1 < d) ? xr[i1] : 0.0;
          p0[s2] = (i0 < d) ? (x0 - n.state_mean[i0]) / n.state_std[i0] : (i0 == d ? pa[0] : (i0 == d + 1 ? pa[1] : 0.0));
          p1[s2] = (i1 < d) ? (x1 - n.state_mean[i1]) / n.state_std[i1] : (i1 == d ? pa[0] : (i1 == d + 1 ? pa[1] : 0.0));
        }
        repfunc_block_mlp<NT3, NS>(n, p0, p1, rows_here, a.slot, fre, fim, &H1[0][0], &H2[0][0], wave, lane);
        __syncthreads();  // F is complete
      }
      // ---- phase B: the QD table, one lane per (dim, sample) row: row rho = 64 slot + lane -> dim rho / SPB, sample rho % SPB.
      // NS = 4: slot = wave = dim.  NS = 2: two dims per wavefront, and the slots rotate over the waves (SIMDs) with the step and
      // the block so that the CU's two workgroups do not pile their QD passes on the same SIMDs.
      {
        const int rot = NS == 4 ? 0 : (int)((t_h + blk) & (NW - 1));
        const int slot_w = (wave + NW - rot) & (NW - 1);
        const int rho = 64 * slot_w + lane;
        const int dim = rho / SPB, sm = rho % SPB;
        if (64 * slot_w < d * SPB && (a.phases & 2)) {  // (wave-uniform: this wavefront holds rows)
          const bool row_ok = dim < d;
          const int dimc = row_ok ? dim : d - 1;
          const int smc = sm < rows_here ? sm : 0;
          cplx res;
          if constexpr (NS == 4) {
            DehoogSlotTerms<CH> src{fre, fim, a.eidx + slot_w * S, (int64_t)SPB, (int64_t)smc, {}};
            res = dehoog_row<M, CH, kDehoogSkewAlone>(src, z);
          } else {
            DehoogSlotTermsLane<CH> src{fre, fim, a.eidx + dimc * S, (int64_t)SPB, (int64_t)smc, {}};
            res = dehoog_row<M, CH, kDehoogSkewAlone>(src, z);
          }
          if (row_ok) {
            // dx is a ROUNDED product, as the staged path's ilt_dehoog_kernel stores it, and the update a separate addition
            // (left contractable, x + (e^{gamma t} / T) Re(res) becomes one fused multiply-add: states off by an ulp per step)
#pragma clang fp contract(off)
            const double dx = exp(gamma * t) / Tt * res.re;
            XS[sm * NLC_MAX_D + dim] = XS[sm * NLC_MAX_D + dim] + dx;  // mppi_with_model.py:120-121
          }
        }
      }
      __syncthreads();
      // ---- tail of the step (wave 0; the other waves go on to the next step's phase A, which only reads XS)
      if (wave == 0 && lane < rows_here) {
        const int64_t k = k0 + lane;
        double x[NLC_MAX_D];
#pragma unroll
        for (int i = 0; i < NLC_MAX_D; ++i) x[i] = XS[lane * NLC_MAX_D + i];
        if (a.states != nullptr)
          for (int i = 0; i < d; ++i) a.states[(k * a.T + t_h) * d + i] = x[i];
        double u[NLC_MAX_NU] = {0.0, 0.0};
        for (int j = 0; j < a.nu; ++j) u[j] = a.u_scale * a.perturbed[(k * a.T + t_h) * a.nu + j];
        const double pc = perturbation_cost_step(a.noise + (k * a.T + t_h) * a.nu, a.U + t_h * a.nu, a.sigma_inv, a.lambda_, a.nu,
                                                 a.noise_abs_cost);
        cost = cost + running_cost(a.env, x, u, a.nu);
        pcost = pcost + pc;
      }
    }
    if (wave == 0 && lane < rows_here) a.cost_total[k0 + lane] = cost + pcost;
    __syncthreads();  // XS is rewritten for the next block
  }
}

// block_tiles: 4 = eight waves per 64 samples (one workgroup per CU), 2 = four waves per 32 samples (two per CU)
hipError_t launch_nl_dehoog_chain(const DehoogChainArgs& a, int block_tiles, hipStream_t s) {
  if (a.K <= 0) return hipSuccess;
  if (a.net.h != 128 || (block_tiles != 2 && block_tiles != 4)) return hipErrorInvalidValue;
  const int64_t spb = 16 * block_tiles, nblk = (a.K + spb - 1) / spb;
  const unsigned grid = (unsigned)(nblk < (1 << 20) ? nblk : (1 << 20));
#define NLC_CHAIN(N, MM)                                                                               \
  if (a.net.nt3 == N && a.net.S == 2 * MM + 1) {                                                       \
    if (block_tiles == 4) {                                                                            \
      hipLaunchKernelGGL((nl_dehoog_chain_kernel<N, MM, 4>), dim3(grid), dim3(512), 0, s, a);          \
    } else {                                                                                           \
      hipLaunchKernelGGL((nl_dehoog_chain_kernel<N, MM, 2>), dim3(grid), dim3(256), 0, s, a);          \
    }                                                                                                  \
    return hipGetLastError();                                                                          \
  }
  // 33 terms (BASELINE configs[4]) and 17 terms (the reference's default count), every state dim 3 .. 6
  NLC_CHAIN(13, 16) NLC_CHAIN(17, 16) NLC_CHAIN(21, 16) NLC_CHAIN(25, 16)
  NLC_CHAIN(7, 8) NLC_CHAIN(9, 8) NLC_CHAIN(11, 8) NLC_CHAIN(13, 8)
#undef NLC_CHAIN
  return hipErrorInvalidValue;
}

bool nl_dehoog_chain_available(int h, int nt3, int S) {
  if (h != 128) return false;
  if (S == 33) return nt3 == 13 || nt3 == 17 || nt3 == 21 || nt3 == 25;
  if (S == 17) return nt3 == 7 || nt3 == 9 || nt3 == 11 || nt3 == 13;
  return false;
}

}  // namespace nlc

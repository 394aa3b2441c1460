// Backward of the de Hoog, Knight & Stokes ILT with respect to the representation-function outputs (theta, phi): the
// reference trains through torchlaplace.laplace_reconstruct with whichever ilt_algorithm its config names
// (train_utils.py:388-407 -> w_nl.py:137-144).  Reverse mode through the quotient-difference table of the forward kernel
// (kernels_dehoog.hip; mpmath 1.3.0 calculus/inverselaplace.py:476-531), one thread per (point, dim) row.
//
// The forward keeps ONE diagonal of the table in registers; reverse mode needs every entry again -- M (M + 1) q's and M^2
// e's, 8.4 KB per row at M = 16 -- so this kernel rebuilds the table column by column (the mpmath column sweep: same rhombus
// rules, same operands per entry as the forward's diagonal sweep) and writes every entry ONCE to a value TAPE in HBM scratch,
// then walks the columns back, reading every value ONCE.  Round 4: the term count is a template parameter, every loop is
// unrolled and every register array is statically indexed, so
//   * the forward sweep keeps its two live columns (q_r and e_(r-1)) in registers and only STORES to the tape (round 3 read
//     the previous column back from it),
//   * the backward sweep keeps its two live ADJOINT columns (qbar_(r+1) -> qbar_r in place, the parked ebar's) in registers
//     (round 3 kept them in a second 9 KB-per-row region of the scratch: 61.5 KB of HBM traffic per row for a 19 KB tape),
//   * the continued-fraction adjoint recurrence runs lazily, two steps before each column, so its seeds dbar_i never leave
//     registers either.
// What travels per row: the value tape (10 KB written + 10 KB read at M = 16), theta / phi in, both gradients out -- 28.2 KB per
// row at the memory counters in round 4, i.e. 92 GB for the bench's 3.3 M rows in 19.8-22.2 ms = 4.2-4.7 TB/s, which is what a
// read + write stream attains on this chip (tools/ubench_hbm_copy.hip: 4.5-5.5): at 33 terms the kernel is bound by its own tape.
// Round 5 took the 8 KB that were NOT tape out of that figure: the rows' inputs and gradients go through an LDS image of the
// wavefront's 64 rows as whole lines (NLC_DHB_LDS_IO below; before, 64 lines per load instruction and eight useful bytes per
// 32-byte sector of every gradient store), and the terms a_k are rebuilt in the epilogue instead of taped: forward + backward
// 22.3 -> 17.1 ms at 3.3 M rows; then the e columns of odd r are rebuilt from their even neighbours instead of taped
// (NLC_DHB_SKIP_ODD_E: 136 of 594 entries not written, and 63 fewer spilled VGPRs): **14.5-15.1 ms**, 0.78 -> 0.58 ms at 81 920
// rows (same box, interleaved; gradients bit-identical to round 4 at every step; the epilogue rebuilds q_1^(k) = a_(k+1) / a_k
// instead of reading column 1 a second time).  Checkpointed columns with recomputation do
// not pay on top of that: DESIGN.md section 9b.
// The scratch belongs to the launch: the grid is persistent, each workgroup (one wavefront, 64 rows) owns one slab, entries
// are [entry][lane] so every access is one 1-KB line per wavefront.
//
// Adjoint convention: for a real loss L and a complex intermediate w, wbar = dL/dRe(w) + i dL/dIm(w); then for
// holomorphic w = f(u): ubar += wbar conj(f'(u)).
#include "nlc_cplx.h"
#include "nlc_device.h"
#include "nlc_kernels.h"

// sweep iterations per prefetch chunk of the backward pass (two chunks of tape values are register-resident; measured on the
// MI355X at 3.3 M rows, M = 16: 8 -> 23.5 ms with 197 spilled VGPRs, 4 -> 21.5 ms with 93, no prefetch 25.6 ms)
#ifndef NLC_DHB_GROUP
#define NLC_DHB_GROUP 4
#endif
// Round 5: the rows' inputs and gradients travel through an LDS image of the wavefront's 64 rows -- (64, S) doubles each for
// theta and phi, read from and written to HBM as whole contiguous lines (a wavefront's 64 consecutive rows ARE one contiguous
// block of 64 S doubles); the gradients overwrite the image in place (term k of a row is read, then written, by the same lane)
// and leave as whole lines.  Before, every lane read and wrote its own row with a stride of S doubles: 64 lines per instruction,
// eight useful bytes per 32-byte sector on the way out.  And the terms a_k are not taped any more: the epilogue rebuilds them from
// theta / phi (the same three operations), 2 x 33 entries = 1 KB per row less on the tape.  0 = the round-4 form.
#ifndef NLC_DHB_LDS_IO
#define NLC_DHB_LDS_IO 1
#endif
// Round 5: the e columns of ODD r are not taped (except their first entry, the continued fraction's coefficient): the backward
// sweep rebuilds e_r^(i) = (q_r^(i+1) - q_r^(i)) + e_(r-1)^(i+1) from the q column it loads anyway and the EVEN column below it
// (zero below column 1) -- the forward's own three operands in the forward's own order, so the rebuilt values are the taped ones
// bit for bit.  136 of the 594 entries per row are not written (2.1 KB), column 1 needs no e loads at all.  0 = tape every column.
#ifndef NLC_DHB_SKIP_ODD_E
#define NLC_DHB_SKIP_ODD_E 1
#endif

namespace nlc {

namespace {

struct DhLayout {
  int M;
  // Round 5: with the defaults (NLC_DHB_LDS_IO, NLC_DHB_SKIP_ODD_E) the terms a_k and the odd e columns are not taped, and the slab
  // is laid out without them: q columns | even e columns | first entry of every odd e column | A | B -- 466 entries per row at M = 16
  // instead of 627 (0.49 GB of scratch for 1024 resident wavefronts instead of 0.66).
  static constexpr bool kCompact = NLC_DHB_LDS_IO && NLC_DHB_SKIP_ODD_E;
  // value tape
  __host__ __device__ constexpr int q(int r, int i) const { return (r - 1) * (2 * M + 2 - r) + i; }                 // r = 1..M, i = 0..2(M-r)+1
  __host__ __device__ constexpr int even_before(int n) const { return n * (2 * M + 1) - 2 * n * (n + 1); }            // entries of the even columns 2 .. 2n
  __host__ __device__ constexpr int e(int r, int i) const {                                                          // r = 1..M, i = 0..2(M-r)
    if (!kCompact) return M * (M + 1) + (r - 1) * (2 * M + 1 - r) + i;
    if ((r & 1) == 0) return M * (M + 1) + even_before((r - 2) / 2) + i;
    return M * (M + 1) + even_before(M / 2) + (r - 1) / 2;  // odd column: only i == 0 is taped
  }
  __host__ __device__ constexpr int e_end() const { return kCompact ? M * (M + 1) + even_before(M / 2) + (M + 1) / 2 : M * (M + 1) + M * M; }
  __host__ __device__ constexpr int a_n() const { return kCompact ? 0 : 2 * M + 1; }
  __host__ __device__ constexpr int a(int i) const { return e_end() + i; }                                            // i = 0..2M (not compact)
  __host__ __device__ constexpr int A(int i) const { return e_end() + a_n() + (i + 1); }                             // i = -1..2M-1
  __host__ __device__ constexpr int B(int i) const { return e_end() + a_n() + (2 * M + 1) + (i + 1); }               // i = -1..2M-1
  __host__ __device__ constexpr int entries() const { return e_end() + a_n() + 2 * (2 * M + 1); }
};
static_assert(!DhLayout::kCompact || DhLayout{16}.entries() == 466, "compact tape layout at M = 16");
static_assert(!DhLayout::kCompact || (DhLayout{16}.e(4, 0) - DhLayout{16}.e(2, 0) == 2 * 14 + 1 && DhLayout{16}.e(16, 0) + 1 == DhLayout{16}.e(1, 0) &&
                                      DhLayout{5}.e(5, 0) + 1 == DhLayout{5}.A(-1) && DhLayout{1}.entries() == 2 + 0 + 1 + 6),
              "compact tape layout: columns are contiguous and disjoint");

// Tape accesses go through a wave-uniform (SGPR) slab pointer that is laundered at every access: the entry offsets are
// compile-time constants of the unrolled sweeps, and left alone the compiler hoists ~1300 per-lane 64-bit addresses out of the
// persistent block loop and spills them (6.8 KB of scratch per lane at M = 16); behind the asm statement every access is
// `global_load / store_dwordx4 v, v_lane_offset, s[base + const]`.
// (a slab entry is (re, im) = 16 bytes per lane; the two scalar accesses below merge into one dwordx4)
typedef __attribute__((address_space(1))) double* tape_ptr;
constexpr int64_t kTapeElemBytes = 16;
struct Tape {
  tape_ptr base;  // this wavefront's slab, [entry][lane][2]; wave-uniform
  int lane;
  __device__ __forceinline__ tape_ptr at(int e) const {
    tape_ptr p = base + (size_t)e * 128;
    asm volatile("" : "+s"(p));
    return p;
  }
  __device__ __forceinline__ cplx ld(int e) const {
    const tape_ptr p = at(e);
    return {p[2 * lane], p[2 * lane + 1]};
  }
  __device__ __forceinline__ void st(int e, cplx v) const {
    const tape_ptr p = at(e);
    p[2 * lane] = v.re;
    p[2 * lane + 1] = v.im;
  }
};

}  // namespace

// (M <= 8: 221 VGPRs, two wavefronts per SIMD; above: up to 360, one)
template <int M>
__global__ __launch_bounds__(64, M <= 8 ? 2 : 1) void ilt_dehoog_bwd_kernel(const IltDehoogBwdArgs a) {
  constexpr int S = 2 * M + 1;
  constexpr DhLayout L{M};
  const int lane = threadIdx.x;
  const int64_t rows_total = a.N * a.d;
  const int64_t nblk = (rows_total + 63) / 64;
  const Tape tp{(tape_ptr)(reinterpret_cast<double*>(a.scratch) + (size_t)blockIdx.x * L.entries() * 128), lane};
  const cplx one = {1.0, 0.0}, zero = {0.0, 0.0};
#if NLC_DHB_LDS_IO
  __shared__ double io_th[64 * S], io_ph[64 * S];  // the wavefront's rows: theta / phi in, the two gradients out (in place)
#endif
  for (int64_t blk = blockIdx.x; blk < nblk; blk += gridDim.x) {
    const int64_t row = blk * 64 + lane;
#if NLC_DHB_LDS_IO
    // one wavefront = one workgroup: the image is private to it, program order + s_waitcnt order its accesses (no barrier)
    const int64_t base = blk * 64 * S;
    // (both copy loops have a WAVE-UNIFORM trip count and a per-lane predicate inside.  A per-lane loop condition -- `for (i = lane;
    // i < n; i += 64)` -- is compiled into a loop that runs until EXEC is empty, and ROCm 7.2 placed the reload of a spilled loop
    // invariant of the persistent block loop behind the second copy loop, BEFORE the point where EXEC is restored: the reload was
    // skipped, the second block of a wavefront divided by garbage and read a.t[] out of bounds -- a memory fault from 1025 blocks
    // on, invisible to every test with at most 1024.  Found in the ISA: `scratch_load ... ; Folded Reload` at a label reached with
    // EXEC = 0.)
    const int live = (int)(rows_total - blk * 64 < 64 ? rows_total - blk * 64 : 64);  // rows of this block (uniform)
    const int n_io = live * S;
    for (int i0 = 0; i0 < n_io; i0 += 64) {
      const int i = i0 + lane;
      if (i < n_io) {
        io_th[i] = a.theta[base + i];
        io_ph[i] = a.phi[base + i];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
#endif
    if (row >= rows_total) continue;  // (no barriers in this kernel: a lane may skip)
    const double t = a.t[row / a.d] / a.t_div;
    const double Tt = a.scale * t;
    const double gamma = a.alpha - a.log_tol / (a.scale * Tt);
    const double ang = kPi * (t / Tt);
    const cplx z = {cos(ang), sin(ang)};
#if NLC_DHB_LDS_IO
    const double* th = io_th + lane * S;
    const double* ph = io_ph + lane * S;
#else
    const double* th = a.theta + row * S;
    const double* ph = a.phi + row * S;
#endif

    // ---------------------------------------------------------------- forward, taped
    // a_k = F_k = R e^{i theta}, R = tan(phi/2 + pi/4); a_0 enters halved.  Column 1: q_1^(i) = a_{i+1} / a_i
    cplx Q[2 * M], E[2 * M];  // the live columns: Q[i] = q_r^(i), E[i] = e_(r-1)^(i)
    cplx a0 = zero;
    {
      cplx prev = zero;
#pragma unroll
      for (int k = 0; k <= 2 * M; ++k) {
        const double rad = m::tan_0_halfpi(ph[k] / 2.0 + kPi / 4.0);
        double sn, cs;
        m::sincos_bounded(th[k], &sn, &cs);
        cplx ak = {rad * cs, rad * sn};
        if (k == 0) ak = cscale(ak, 0.5);
        if (k == 0) a0 = ak;
        if (!NLC_DHB_LDS_IO) tp.st(L.a(k), ak);
        if (k > 0) {
          Q[k - 1] = cdiv(ak, prev);
          tp.st(L.q(1, k - 1), Q[k - 1]);
        }
        prev = ak;
      }
    }
#pragma unroll
    for (int i = 0; i < 2 * M; ++i) E[i] = zero;
    // continued fraction: d_0 = a_0, d_(2r-1) = -q_r^(0), d_(2r) = -e_r^(0);  A_i = A_(i-1) + d_i z A_(i-2), fed as the columns
    // produce the coefficients
    cplx A_prev = zero, A_cur = a0, B_prev = one, B_cur = one;
    tp.st(L.A(-1), A_prev);
    tp.st(L.A(0), A_cur);
    tp.st(L.B(-1), B_prev);
    tp.st(L.B(0), B_cur);
    cplx d_last = zero, d_end = zero;
    auto feed = [&](int i, cplx di) {  // i = 1 .. 2M
      d_last = d_end;
      d_end = di;
      if (i <= 2 * M - 1) {
        const cplx dz = cmul(di, z);
        const cplx An = cadd(A_cur, cmul(dz, A_prev)), Bn = cadd(B_cur, cmul(dz, B_prev));
        A_prev = A_cur;
        A_cur = An;
        B_prev = B_cur;
        B_cur = Bn;
        tp.st(L.A(i), A_cur);
        tp.st(L.B(i), B_cur);
      }
    };
#pragma unroll
    for (int r = 1; r <= M; ++r) {
      const int mr = 2 * (M - r) + 1;
      feed(2 * r - 1, cneg(Q[0]));
      // one pass up the column, both live columns updated in place:
      //   e_r^(i)       = q_r^(i+1) - q_r^(i) + e_(r-1)^(i+1)
      //   q_(r+1)^(i-1) = q_r^(i) e_r^(i) / e_r^(i-1)                      (r < M, i >= 1)
      cplx qlo = Q[0], elo = zero;
#pragma unroll
      for (int i = 0; i < mr; ++i) {
        const cplx qhi = Q[i + 1];
        const cplx ei = cadd(csub(qhi, qlo), E[i + 1]);
        if (!NLC_DHB_SKIP_ODD_E || (r & 1) == 0 || i == 0) tp.st(L.e(r, i), ei);
        if (r != M && i >= 1) {
          Q[i - 1] = cdiv(cmul(qlo, ei), elo);
          tp.st(L.q(r + 1, i - 1), Q[i - 1]);
        }
        E[i] = ei;
        qlo = qhi;
        elo = ei;
        if ((i & (NLC_DHB_GROUP - 1)) == NLC_DHB_GROUP - 1) __builtin_amdgcn_sched_barrier(0);
      }
      feed(2 * r, cneg(E[0]));
      // (keeps the columns apart: the compiler otherwise interleaves them and the live ranges explode)
      __builtin_amdgcn_sched_barrier(0);
    }
    const cplx brem = cscale(cadd(one, cmul(csub(d_last, d_end), z)), 0.5);
    const cplx uoverb = cdiv(cmul(d_end, z), brem);  // inner - 1
    const cplx sq = csqrt_(cadd(one, uoverb));
    const cplx sm1 = csub(sq, one);
    const cplx rem = cmul(brem, sm1);
    const cplx An = cadd(A_cur, cmul(rem, A_prev));
    const cplx Bn = cadd(B_cur, cmul(rem, B_prev));
    const cplx res = cdiv(An, Bn);

    // ---------------------------------------------------------------- backward
    const double G = a.gx[row] * (exp(gamma * t) / Tt);  // x = e^{gamma t} / T Re(res)
    const cplx g_res = {G, 0.0};
    // res = An / Bn
    const cplx g_An = cdiv(g_res, cconj(Bn));
    const cplx g_Bn = cneg(cmul(g_An, cconj(res)));
    // An = A_cur + rem A_prev,  Bn = B_cur + rem B_prev
    cplx gA1 = g_An, gA0 = cmul(g_An, cconj(rem));  // adjoints of (A_(2M-1), A_(2M-2))
    cplx gB1 = g_Bn, gB0 = cmul(g_Bn, cconj(rem));
    const cplx g_rem = cadd(cmul(g_An, cconj(A_prev)), cmul(g_Bn, cconj(B_prev)));
    // rem = brem (sq - 1);  sq = sqrt(inner);  inner = 1 + (d_end z) / brem
    cplx g_brem = cmul(g_rem, cconj(sm1));
    const cplx g_sq = cmul(g_rem, cconj(brem));
    const cplx g_inner = cdiv(g_sq, cscale(cconj(sq), 2.0));
    const cplx g_u = cdiv(g_inner, cconj(brem));
    g_brem = csub(g_brem, cmul(g_inner, cconj(cdiv(uoverb, brem))));
    cplx g_dend = cmul(g_u, cconj(z));
    // brem = (1 + (d_last - d_end) z) / 2
    const cplx g_diff = cscale(cmul(g_brem, cconj(z)), 0.5);
    const cplx g_dlast = g_diff;
    g_dend = csub(g_dend, g_diff);
    // one step of the recurrence's adjoint, i = 2M-1 .. 1 (called in descending order): returns dbar_i.  Its three tape operands
    // (d_i's table entry, A_(i-2), B_(i-2)) are loaded a column AHEAD of their use (UnfeedOps), like every other tape value of
    // the sweep: the kernel runs one wavefront per SIMD, so nothing else hides a load's latency.
    struct UnfeedOps {
      cplx entry, Am2, Bm2;
    };
    auto unfeed_load = [&](int i) -> UnfeedOps {
      return {(i & 1) ? tp.ld(L.q((i + 1) / 2, 0)) : tp.ld(L.e(i / 2, 0)), tp.ld(L.A(i - 2)), tp.ld(L.B(i - 2))};
    };
    auto unfeed = [&](int i, const UnfeedOps& o) -> cplx {
      const cplx di = cneg(o.entry);
      cplx g_di = cadd(cmul(gA1, cconj(cmul(z, o.Am2))), cmul(gB1, cconj(cmul(z, o.Bm2))));
      if (i == 2 * M - 1) g_di = cadd(g_di, g_dlast);
      const cplx cdz = cconj(cmul(di, z));
      const cplx gAm2 = cmul(gA1, cdz), gBm2 = cmul(gB1, cdz);
      gA0 = cadd(gA0, gA1);
      gB0 = cadd(gB0, gB1);
      gA1 = gA0;
      gA0 = gAm2;
      gB1 = gB0;
      gB0 = gBm2;
      return g_di;
    };
    // table, columns r = M .. 1.  With wbar_j = qbar_(r+1)^(j) (zero for r = M):
    //   ebar_r^(i) = [i = 0] (-dbar_2r) + ebar_(r+1)^(i-1)                                    (e_(r+1)^(i-1) = ... + e_r^(i))
    //               + wbar_(i-1) conj(q_r^(i) / e_r^(i-1)) - wbar_i conj(q_(r+1)^(i) / e_r^(i))   (q_(r+1) = q e_hi / e_lo)
    //   qbar_r^(i) = [i = 0] (-dbar_(2r-1)) + ebar_r^(i-1) - ebar_r^(i) + wbar_(i-1) conj(e_r^(i) / e_r^(i-1))
    // W[i]: qbar_(r+1)^(i) on entry to column r, qbar_r^(i) on exit (in place).  P[i]: ebar_(r+1)^(i-1), parked by the sweep of
    // column r + 1 in the slot of ebar_r^(i); the sweep of column r parks ebar_r^(i) in P[i + 1].
    cplx W[2 * M], P[2 * M];
#pragma unroll
    for (int i = 0; i < 2 * M; ++i) {
      W[i] = zero;
      P[i] = zero;
    }
    // Tape values are fetched in chunks of kChunk sweep iterations, one chunk AHEAD of the chunk being computed (two register
    // buffers), and the next column's recurrence operands one column ahead.
    constexpr int kChunk = NLC_DHB_GROUP;
    UnfeedOps uo_even = unfeed_load(2 * M - 1), uo_odd = uo_even;  // column M: dbar_2M is g_dend, dbar_(2M-1) needs step 2M-1
#pragma unroll
    for (int r = M; r >= 1; --r) {
      const int mr = 2 * (M - r) + 1;
      const bool inner = r != M;
      const int nch = (mr + 1 + kChunk - 1) / kChunk;
      cplx qn[2][kChunk], en[2][kChunk];  // qn[b][j] = q_r^(i+1), en[b][j] = e_r^(i+1) for sweep iteration i = c kChunk + j
      // first chunk + the sweep's first entries
      cplx e_i = tp.ld(L.e(r, 0)), q_i = tp.ld(L.q(r, 0));
      // odd columns (NLC_DHB_SKIP_ODD_E): e_r is rebuilt, so the q buffers run TWO entries ahead (qn = q_r^(i+2), q_roll =
      // q_r^(i+1)) and the e buffers hold the even column below, en = e_(r-1)^(i+2)
      const bool odd = NLC_DHB_SKIP_ODD_E && (r & 1) != 0;
      cplx q_roll = (odd && 1 <= mr) ? tp.ld(L.q(r, 1)) : zero;
      auto load_q = [&](int i) -> cplx {  // the q entry iteration i's buffer slot holds
        const int at = odd ? i + 2 : i + 1;
        return (i <= mr && at <= mr) ? tp.ld(L.q(r, at)) : zero;
      };
      auto load_e = [&](int i) -> cplx {  // ... and the e entry (even: e_r^(i+1); odd: e_(r-1)^(i+2), zero below column 1)
        if (!(i <= mr && i + 1 <= mr - 1)) return one;
        if (!odd) return tp.ld(L.e(r, i + 1));
        return r > 1 ? tp.ld(L.e(r - 1, i + 2)) : zero;
      };
#pragma unroll
      for (int j = 0; j < kChunk; ++j) {
        qn[0][j] = load_q(j);
        en[0][j] = load_e(j);
      }
      // the recurrence steps of THIS column (operands loaded during the previous column), then the next column's operands
      const cplx dbar_even = r == M ? g_dend : unfeed(2 * r, uo_even);  // dbar_(2r)
      const cplx dbar_odd = unfeed(2 * r - 1, r == M ? uo_even : uo_odd);  // dbar_(2r-1)
      if (r > 1) {
        uo_even = unfeed_load(2 * (r - 1));
        uo_odd = unfeed_load(2 * (r - 1) - 1);
      }
      cplx g_prev = zero, wbar_im1 = zero, e_im1 = one;
      cplx park_in = zero;  // the old P[i] (read before the sweep overwrites it one iteration earlier)
#pragma unroll
      for (int c = 0; c < nch; ++c) {
        if (c + 1 < nch) {
#pragma unroll
          for (int j = 0; j < kChunk; ++j) {
            const int i = (c + 1) * kChunk + j;
            qn[(c + 1) & 1][j] = load_q(i);
            en[(c + 1) & 1][j] = load_e(i);
          }
        }
        // (the next chunk's loads are issued before this chunk's arithmetic and cannot sink below it)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < kChunk; ++j) {
          const int i = c * kChunk + j;
          if (i <= mr) {
            const bool has_e = i <= mr - 1;
            const bool has_w = inner && i <= mr - 2;  // q_(r+1)^(i) exists
            cplx q_nxt = qn[c & 1][j];
            cplx e_nxt = en[c & 1][j];
            if (odd) {
              const cplx q_nn = q_nxt;  // q_r^(i+2)
              q_nxt = q_roll;           // q_r^(i+1)
              // e_r^(i+1) as the forward built it: (q_r^(i+2) - q_r^(i+1)) + e_(r-1)^(i+2)
              if (i + 1 <= mr - 1) e_nxt = cadd(csub(q_nn, q_nxt), e_nxt);
              q_roll = q_nn;
            }
            const cplx wbar_i = has_w ? W[i] : zero;
            const cplx park_nxt = (i + 1 < 2 * M) ? P[i + 1] : zero;  // old value, before this iteration parks into it
            cplx g = zero;
            cplx c_term = zero;  // wbar_(i-1) conj(e_r^(i) / e_r^(i-1))
            if (has_e) {
              const double inv_i = m::rcp_refined(fma(e_i.re, e_i.re, e_i.im * e_i.im));
              const cplx ie_i = {e_i.re * inv_i, -e_i.im * inv_i};  // 1 / e_r^(i)
              if (i == 0) g = cneg(dbar_even);
              if (inner && i >= 1 && i <= mr - 2) g = cadd(g, park_in);  // parked ebar_(r+1)^(i-1)
              if (inner && i >= 1) {
                const double inv_m = m::rcp_refined(fma(e_im1.re, e_im1.re, e_im1.im * e_im1.im));
                const cplx ie_m = {e_im1.re * inv_m, -e_im1.im * inv_m};  // 1 / e_r^(i-1)
                g = cadd(g, cmul(wbar_im1, cconj(cmul(q_i, ie_m))));
                c_term = cmul(wbar_im1, cconj(cmul(e_i, ie_m)));
              }
              if (has_w) {
                // q_(r+1)^(i) / e_r^(i) = q_r^(i+1) e_r^(i+1) / e_r^(i)^2
                const cplx ratio = cmul(e_nxt, ie_i);
                g = csub(g, cmul(wbar_i, cconj(cmul(cmul(q_nxt, ie_i), ratio))));
              }
            }
            cplx qb = csub(cadd(g_prev, c_term), g);
            if (i == 0) qb = csub(qb, dbar_odd);
            W[i] = qb;
            if (has_e && r > 1 && i + 1 < 2 * M) P[i + 1] = g;
            park_in = park_nxt;
            g_prev = g;
            wbar_im1 = wbar_i;
            e_im1 = e_i;
            e_i = e_nxt;
            q_i = q_nxt;
          }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // A_0 = d_0 = a_0 (A_(-1), B_0, B_(-1) are constants): d_0-bar = A_0-bar
    const cplx g_a0_seed = gA1;
    // column 1 (q_1^(i) = a_(i+1) / a_i) and the sphere map, per term k:
    //   abar_k = [k = 0] dbar_0 + h_(k-1) - h_k conj(q_1^(k)),  h_k = qbar_1^(k) / conj(a_k)  (k <= 2M-1)
    //   F_k = R (cos theta + i sin theta), R = tan(phi/2 + pi/4), dR/dphi = (1 + R^2) / 2;  a_0 = F_0 / 2
    cplx h_prev = zero;
    // (rad, sn, cs) of term k + 1 are computed one iteration ahead: a_(k+1) / a_k IS q_1^(k), by the forward's own division, so
    // column 1 is not read a second time here (NLC_DHB_LDS_IO; the image still holds theta / phi of every term > k)
    double rad_n = m::tan_0_halfpi(ph[0] / 2.0 + kPi / 4.0), sn_n, cs_n;
    m::sincos_bounded(th[0], &sn_n, &cs_n);
#pragma unroll
    for (int k = 0; k <= 2 * M; ++k) {
      const double rad = rad_n, sn = sn_n, cs = cs_n;
      if (k + 1 <= 2 * M) {
        rad_n = m::tan_0_halfpi(ph[k + 1] / 2.0 + kPi / 4.0);
        m::sincos_bounded(th[k + 1], &sn_n, &cs_n);
      }
      cplx gF = k == 0 ? g_a0_seed : h_prev;
      if (k <= 2 * M - 1) {
#if NLC_DHB_LDS_IO
        cplx ak = {rad * cs, rad * sn};  // a_k as the forward built it (same operations)
        if (k == 0) ak = cscale(ak, 0.5);
        const cplx ak1 = {rad_n * cs_n, rad_n * sn_n};
        const cplx q1k = cdiv(ak1, ak);
#else
        const cplx ak = tp.ld(L.a(k));
        const cplx q1k = tp.ld(L.q(1, k));
#endif
        const cplx h = cdiv(W[k], cconj(ak));
        gF = csub(gF, cmul(h, cconj(q1k)));
        h_prev = h;
      }
      if (k == 0) gF = cscale(gF, 0.5);
      const double g_th = rad * (gF.im * cs - gF.re * sn);
      const double g_ph = (gF.re * cs + gF.im * sn) * (0.5 * (1.0 + rad * rad));
#if NLC_DHB_LDS_IO
      io_th[lane * S + k] = g_th;  // in place: term k of this row is not read again
      io_ph[lane * S + k] = g_ph;
#else
      a.gtheta[row * S + k] = g_th;
      a.gphi[row * S + k] = g_ph;
#endif
    }
#if NLC_DHB_LDS_IO
    // (lanes past the last row skipped the body with `continue`: in a partial block they do not reach this point, the live
    // lanes copy the whole image out)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    for (int i0 = 0; i0 < n_io; i0 += live) {  // (the live lanes are 0 .. live - 1)
      const int i = i0 + lane;
      if (i < n_io) {
        a.gtheta[base + i] = io_th[i];
        a.gphi[base + i] = io_ph[i];
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
#endif
  }
}

// The two adjoint columns alone are 256 VGPRs at M = 16: one wavefront per SIMD above M = 8 (accumulation registers as overflow),
// two up to M = 8.  Every resident workgroup (= wavefront) holds a slab: at most 4 / 8 per CU, i.e. 1024 x 642 KB = 0.66 GB at 33
// terms (round 3 took 2.5 GB from the stream-ordered pool: ADVICE r3).
int64_t ilt_dehoog_bwd_scratch_bytes(int64_t N, int d, int S, unsigned* grid_out) {
  const int64_t nblk = (N * d + 63) / 64;
  // (slabs in flight, measured at 3.3 M rows, S = 33: 256 -> 36.3 ms, 512 -> 29.8, 768 -> 22.7, 1024 -> 19.8, 2048 (1024
  // resident) -> 20.9: fewer slabs would fit the Infinity Cache but starve the latency hiding)
  const int64_t cap = (S - 1) / 2 <= 8 ? 2048 : 1024;
  const unsigned grid = (unsigned)(nblk < cap ? nblk : cap);
  if (grid_out) *grid_out = grid;
  const DhLayout L{(S - 1) / 2};
  return (int64_t)grid * L.entries() * 64 * kTapeElemBytes;
}

hipError_t launch_ilt_dehoog_bwd(const IltDehoogBwdArgs& a, hipStream_t s) {
  if (a.N * a.d <= 0) return hipSuccess;
  if (a.S < 3 || a.S > 33 || (a.S & 1) == 0 || !a.scratch) return hipErrorInvalidValue;
  unsigned grid = 0;
  ilt_dehoog_bwd_scratch_bytes(a.N, a.d, a.S, &grid);
  switch ((a.S - 1) / 2) {
#define NLC_DHB(MM)                                                                              \
  case MM:                                                                                       \
    hipLaunchKernelGGL((ilt_dehoog_bwd_kernel<MM>), dim3(grid), dim3(64), 0, s, a);              \
    break;
    NLC_DHB(1) NLC_DHB(2) NLC_DHB(3) NLC_DHB(4) NLC_DHB(5) NLC_DHB(6) NLC_DHB(7) NLC_DHB(8)
    NLC_DHB(9) NLC_DHB(10) NLC_DHB(11) NLC_DHB(12) NLC_DHB(13) NLC_DHB(14) NLC_DHB(15) NLC_DHB(16)
#undef NLC_DHB
    default:
      return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace nlc

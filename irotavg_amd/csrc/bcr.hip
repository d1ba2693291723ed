// bcr.hip -- direct solve of a BANDED level-0 operator (a view sequence without loop closures: every view is
// linked to a bounded number of predecessors) by block cyclic reduction. Replaces the sparse factorisations
// of the reference on such graphs: SuiteSparseQR in irls (ral/l1_irls.cpp:536-556) and UMFPACK in
// l1decode_pd (ral/l1_irls.cpp:131-184) -- no iteration, no preconditioner, no tolerance.
//
// The operator A = A' D^2 A (IRLS) or the primal-dual Hessian (L1RA) of a graph whose edges (i, j) all have
// |i - j| <= b is a symmetric positive definite band matrix of half-bandwidth b. With B >= b rows per block
// it is block tridiagonal: D_i = A[i, i] (B x B), G_i = A[i, i + 1] (B x B). A workgroup takes a CHUNK of
// eight consecutive blocks and eliminates seven of them in nested-dissection order (0 2 4 6 | 1 5 | 3),
// one wave per block, leaving block 7 -- the separator -- coupled to the separator of the chunk before
// it. Eliminating block i between its current neighbours a (left) and c (right):
//     W = D_i^-1 [ P' | Q | R_i ],  P = A[a, i], Q = A[i, c]            (B x (2B + 3), kept for the way back)
//     D_a -= P W_P,  A[a, c] = -P W_Q,  R_a -= P W_R,   D_c -= Q' W_Q,  R_c -= Q' W_R
// The separators form a block tridiagonal system an eighth the size: the same kernel again, until at
// most eight blocks are left (100k views, B = 24: 4167 -> 521 -> 66 -> 9 -> 2 blocks). The way back is
// x_i = W_R - W_P x_a - W_Q x_c level by level. No host round trip anywhere: a solve is ~10 launches.
//
// What the hardware dictated: the B^3 work (five products per block) runs on the matrix cores
// (v_mfma_f64_16x16x4_f64: operands are distributed over the lanes, so no operand is broadcast through
// LDS -- with vector FMAs every value of P, Q and D^-1 would have to reach all 64 lanes, and eight waves
// per CU saturate the LDS port with that); the accumulator layout of one product IS the B-operand layout
// of the next (row = 4 reg + lane / 16), so W never leaves the registers between the two. D_i^-1 is
// formed by a symmetric sweep (Gauss-Jordan without pivoting: SPD), a column per lane, the pivot row
// broadcast through LDS (bcr_invert). A pivot not above kDeadTol x its original diagonal entry is dead
// (an isolated view, a floating component): its unknown solves to 0, as everywhere else in this library.
#include <atomic>

#include "graph.hpp"
#include "kernels.hpp"

namespace irh {
// a direct solve with closures is repeated by conjugate gradients on the full operator when its relative residual is above
// the tolerance the ITERATIVE solver would have been held to (options.pcg_rtol, 1e-10 by default; at least this). A sound
// Woodbury solve: 1e-14 ... 2e-11 with a thousand closures at 100k views; what slipped through looser gates: fuzz seed
// 21 case 35 at 8e-9 (1.3e-6 rad off the oracle), seed 32 case 1 below 1e-9 (3e-8 rad).
constexpr double kBcrGateTolMin = 1e-12;

typedef double v4d __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));
typedef int v2i __attribute__((ext_vector_type(2)));

constexpr int kTopMax = 16, kTopRounds = 6;
struct BcrTopSched {
    int n = 0, nr = 0, last = 0;   // blocks, rounds, the block that is left
    signed char ne[kTopRounds];    // eliminations of a round (<= 4)
    signed char i[kTopRounds][4], a[kTopRounds][4], c[kTopRounds][4];  // block, left / right neighbour (-1: none)
};

// (the arguments of k_bcr_reduce_up, see there). They live in device memory and are read by functions that are not inlined
// into the kernel: a pointer loaded from memory is GENERIC to the compiler (flat loads / stores, which also count on the
// LDS counter) unless its type says global -- so the fields are typed global for the device pass.
#if defined(__HIP_DEVICE_COMPILE__)
#define IRH_GPTR(T) T __attribute__((address_space(1))) *
#else
#define IRH_GPTR(T) T *
#endif
typedef IRH_GPTR(const double) gp_cd;
typedef IRH_GPTR(double) gp_d;
typedef IRH_GPTR(const int) gp_ci;
typedef IRH_GPTR(int) gp_i;
typedef IRH_GPTR(unsigned) gp_u;
typedef IRH_GPTR(const double4) gp_cd4;
constexpr int kUpLevels = 8;
struct BcrUpLevel {
    int nb, nred, nch;
    gp_cd inD, inR, inXD, inXR, inXG;  // from the level below
    gp_d W, sepD, sepR, extD, extR, extG;
};
struct BcrUpArgs {
    int nl;  // levels of the launch; lev[0] = level 1 of the solve
    BcrUpLevel lev[kUpLevels];
    // the level-0 operator, for the blocks of a mixed level 1 that no chunk reduced
    int n;
    gp_ci sl_off, col;
    gp_cd val, diag;
    gp_cd4 rhs;
    gp_d xtop;
    gp_u cnt;
    // round 5: the last level as ONE workgroup of up to sixteen blocks that also makes its way back (bcr_top_body):
    // top16 != 0, its schedule, where the solutions of its blocks go
    int top16;
    BcrTopSched top;
    gp_d xtop_all;
    // a wait that did not end: every workgroup that gives up says so here (cnt + kUpLevels), skips its chunk but still
    // counts itself -- nobody else waits a second for it --, and the top workgroup poisons the solution when the word is
    // set; the host then repeats the solve level by level (bcr_up_failed)
    gp_i fail;
};

struct BcrLevel {
    int nb = 0;   // blocks
    int nred = 0; // ... of which the first nred come from the level below (the others: raw level-0 blocks)
    int nch = 0;  // chunks of eight
    DevBuf<double> W;                             // nch * 7 blocks of B x (2B + 3)
    DevBuf<double> sepD, sepR, extD, extR, extG;  // per chunk: what the next level is assembled from
    DevBuf<double> x;                             // nch * 8 blocks of B x 3 (levels >= 1)
};
struct BcrState {
    int B = 0;
    int NR = 3;    // right-hand-side columns riding along: 3, or 19 with closures
    int nfar = 0;  // long-range edges (loop closures) handled by the Woodbury correction
    std::vector<BcrLevel> lev;
    DevBuf<double> xtop;  // B x NR: the last separator
    DevBuf<int> far_i, far_j, far_e;  // rows and edge id of the closures
    // the closures' part of a solve (bcr_closures): per level the inverse of every eliminated block, the step programs
    // of the closures' forward eliminations and what they record, the Woodbury system
    std::vector<DevBuf<double>> Dinv;
    DevBuf<double> topDinv;
    DevBuf<int> cl_off;          // nfar + 1: first step of a closure
    DevBuf<int4> cl_step;        // {level (-1: the top block), elimination inside the level, slots src | dstA << 8 | dstC << 16, sort key}
    DevBuf<int4> cl_init;        // {slot, row inside the block} of the two endpoints
    DevBuf<int> cl_owner;        // the closure of a step
    DevBuf<double> cl_R, cl_W;   // per step: the block's right-hand side when it was eliminated, and D^-1 times it
    DevBuf<double> cl_S, cl_T, lam;  // Woodbury system (npad x npad, npad x 3), its solution (nfar x 3)
    DevBuf<int> cl_alive;
    DevBuf<int> dead;            // dead pivots of the band factor of the last solve
    DevBuf<int> co_off, co_rec;  // per touched elimination: the steps that touch it ...
    DevBuf<int2> co_elim;        // ... and {level, elimination}
    int cl_nslots = 0, cl_nsteps = 0, cl_nelim = 0, cl_npad = 0, cl_maxsteps = 1, cl_ndense = 0;
    DevBuf<int> cl_dense;        // cl_ndense x nfar: the step of closure q on dense elimination d, or -1
    // a shard of a sharded sequence (dist.hip): the separator before the first chunk is the previous rank's
    int ext0 = 0;
    int dbg_word = -1;         // development aid: the debug word of the NEXT solves of this handle (bcr_stamps*); -1: IROTAVG_BCR_DBG
    DevBuf<long long> stamps;  // development aid: bcr_stamp
    DevBuf<unsigned> up_cnt;   // k_bcr_reduce_up: arrivals per level boundary (they only grow: solve g waits for g x chunks);
                               // word kUpLevels: a wait gave up (bcr_up_failed)
    unsigned up_gen = 0;
    DevBuf<BcrUpArgs> up_args; // k_bcr_reduce_up's arguments (written once)
    bool up_used = false;      // the last solve went through k_bcr_reduce_up
    bool top16 = false;        // the last level is one workgroup of <= 16 blocks that makes its own way back (bcr_top_body)
    BcrTopSched tsched;
    DevBuf<BcrTopSched> tsched_dev;
    DevBuf<int> ghost_extcol;  // per ghost view: its row in the previous rank's last block, or -1
    DevBuf<double> remD, remR; // what this shard's eliminations subtract from that separator (sum over its levels)
    // closures on a shard: per local closure the slots that hold what is left on this rank's separator and on the one
    // before it {own, remote} (255: nothing), its number in the global list, who adds its 1 / w; the rows whose diagonal
    // loses the weight of a closure to a ghost view for the time of the reduction
    DevBuf<double> res_part;   // k_bcr_residual's partial sums (bcr_gate)
    DevBuf<int> res_zero;      // FL_COUNT zero words: the done flags the gate's SpMV runs behind
    DevBuf<int> gate_skip;     // k_bcr_gate's verdict as a skip word (1: do not touch weights / residuals)
    bool dead_clean = false;   // the dead-pivot counter is known to be zero (k_bcr_gate read and cleared it)
    DevBuf<int2> cl_fin;
    DevBuf<int> cl_gid;
    DevBuf<uint8_t> cl_own;
    DevBuf<int> fix_row, fix_off, fix_e;
    DevBuf<double> fix_saved;
    int nfix = 0;
};

void BcrDeleter::operator()(BcrState *p) const { delete p; }

__device__ __forceinline__ double bcr_readlane(double v, int lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, lane);
    hi = __builtin_amdgcn_readlane(hi, lane);
    return __hiloint2double(hi, lo);
}

// Sum over aligned groups of sixteen lanes, every lane gets it: the butterfly of __shfl_xor(1, 2, 4, 8) -- the same
// additions, bit for bit -- by DPP moves (quad permutes, then the mirrors of half a row and of a row: after two steps all
// four lanes of a quad hold the quad's sum, so WHICH lane of the other quad / half a lane adds does not matter) instead
// of ds_bpermute: a step costs ~20 clocks instead of an LDS round trip (round 5: the ways back are chains of such sums).
template <int CTRL>
__device__ __forceinline__ double bcr_dpp(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double bcr_row16_sum(double v) {
    v += bcr_dpp<0xB1>(v);   // quad_perm [1 0 3 2]
    v += bcr_dpp<0x4E>(v);   // quad_perm [2 3 0 1]
    v += bcr_dpp<0x141>(v);  // row_half_mirror
    v += bcr_dpp<0x140>(v);  // row_mirror
    return v;
}

// In-place inverse of the SPD B x B matrix Dm (LDS, row-major) by one wave; B even, B <= 32. Symmetric sweep: after
// pivot k the registers hold -(inverse) on the swept part. Lane (c, h) = (lane % 32, lane / 32) holds the rows
// h HB .. h HB + HB - 1 of column c (HB = B / 2). The sweep stays symmetric, so column k IS row k: per pivot the lanes
// that hold row k write it to LDS (one store) and every lane reads back its own column's entry (the multiplier of the
// pivot row) and the HB entries of its rows (the pivot column) -- broadcast reads, no lane permutes, no address
// arithmetic (round 3 exchanged column k by 12 ds_bpermute on an 8 x 8-lane register tile and needed B % 8 == 0). The
// pivot and its dead-pivot reference travel through SGPRs (v_readlane), so the reciprocal starts before the LDS round
// trip of the row has finished. A wave issues in order -- about one instruction per 4.5 clocks, whatever it is -- so the
// chain of a pivot and the issue time of its updates ADD unless independent work stands between the links: the element
// of row k + 1 is updated first, written, its reciprocal started and the reads of row k + 1 issued; the other updates
// of pivot k follow and cover those latencies. (The empty asm statements pin that right-looking order: without them
// the compiler defers an element's updates until its row is the pivot row -- k dependent FMAs on the critical path.)
// LDS scratch: the first B doubles of the block itself (it lives in registers during the sweep). One wave alone:
// 0.85 / 2.3 / 4.2 / 6.9 us for B = 8 / 16 / 24 / 32 (tools/micro/sweep_cols.hip).
// dead: a counter of dead pivots (nullptr: not wanted); reg: a dead pivot is replaced by the row's own original
// diagonal entry (1 if that is not positive) instead of zeroing the unknown -- the regularised factorisation the guard of
// the closure path preconditions with (Graph::bcr_guard).
template <int B>
__device__ __forceinline__ void bcr_invert(double *Dm, int lane, int *dead = nullptr, bool reg = false) {
    static_assert(B % 2 == 0 && B <= 32, "block size");
    constexpr int HB = B / 2;
    const int c = lane & 31, h = lane >> 5;
    const bool act = c < B;
    const int cl = act ? c : B - 1;
    const v2d *Dh = reinterpret_cast<const v2d *>(Dm + h * HB);
    double t[HB];
#pragma unroll
    for (int i = 0; i < HB; i++) t[i] = Dm[(h * HB + i) * B + cl];
    const double dg = Dm[cl * B + cl];
    const double dgt = kDeadTol * dg;  // (the dead-pivot limit of the lane's row, formed once: a product per pivot less)
    int ndead = 0;
    auto pivot_inverse = [&](double tk, int k, int kh) {
        // reciprocal by v_rcp_f64 + two Newton steps (the IEEE division sequence is three times as long and sits on
        // the chain from pivot to pivot)
        const double p0 = bcr_readlane(tk, k + 32 * kh);
        const bool alive = p0 > bcr_readlane(dgt, k);
        ndead += alive ? 0 : 1;
        double p = p0;
        if (reg && !alive) {
            const double ref = bcr_readlane(dg, k);
            p = ref > 0.0 ? ref : 1.0;
        }
        double x = __builtin_amdgcn_rcp(p);
        x = fma(fma(-p, x, 1.0), x, x);
        x = fma(fma(-p, x, 1.0), x, x);
        return (alive || reg) ? x : 0.0;
    };
    // (the row's own diagonal position carries -1 instead of the pivot: nobody reads the pivot there -- it travels through
    // an SGPR --, and the lane of column k, which reads it as its multiplier of the pivot row, gets rowk pinv = -pinv, the
    // (k, k) entry of the swept matrix, without a select of its own)
    if (h == 0 && act) Dm[c] = c == 0 ? -1.0 : t[0];
    asm volatile("" ::: "memory");
    double pinv = pivot_inverse(t[0], 0, 0);
    double rowk = Dm[cl], colk[HB];
#pragma unroll
    for (int i = 0; i < HB; i += 2) {
        const v2d v = Dh[i / 2];
        colk[i] = v.x;
        colk[i + 1] = v.y;
    }
#pragma unroll
    for (int k = 0; k < B; k++) {
        const int kh = k / HB, ki = k - kh * HB;
        const int k1 = k + 1, kh1 = k1 / HB, ki1 = k1 - kh1 * HB;  // the next pivot (when k1 < B)
        const bool isk = c == k;
        const double f = rowk * pinv;
        const double g = f;
        // general element: t -= colk (rowk pinv); column k: colk pinv; row k: rowk pinv; (k, k): -pinv
        // (the lane of column k drops its old values by a product with 0, one instruction, not by a 64-bit select, two:
        // 24 of the ~105 instructions of a pivot; taking the LDS round trip of the row out of the pivot-to-pivot chain --
        // the element of row k + 1 updated from registers alone, its multiplier through an SGPR like the pivot -- cost ten
        // instructions per pivot and made the sweep slower, 11 600 instead of 10 150 clocks: the chain is the reciprocal's)
        const double keep = isk ? 0.0 : 1.0;
        auto upd = [&](int i) {
            const double a = t[i] * keep;
            double u = fma(-colk[i], g, a);
            if (i == ki) u = (h == kh) ? g : u;
            t[i] = u;
        };
        double pinv1 = 0.0, rowk1 = 0.0, colk1[HB];
        if (k1 < B) {
            upd(ki1);
            if (h == kh1 && act) Dm[c] = c == k1 ? -1.0 : t[ki1];
            asm volatile("" ::: "memory");
            pinv1 = pivot_inverse(t[ki1], k1, kh1);
            rowk1 = Dm[cl];
#pragma unroll
            for (int i = 0; i < HB; i += 2) {
                const v2d v = Dh[i / 2];
                colk1[i] = v.x;
                colk1[i + 1] = v.y;
            }
        }
#pragma unroll
        for (int i = 0; i < HB; i++)
            if (!(k1 < B && i == ki1)) upd(i);
#pragma unroll
        for (int i = 0; i < HB; i++) asm volatile("" : "+v"(t[i]));
        if (k1 < B) {
            pinv = pinv1;
            rowk = rowk1;
#pragma unroll
            for (int i = 0; i < HB; i++) colk[i] = colk1[i];
        }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int i = 0; i < HB; i++)
        if (act) Dm[(h * HB + i) * B + c] = -t[i];
    if (dead && ndead && lane == 0) atomicAdd(dead, ndead);
}

// development aid (IROTAVG_BCR_DBG & 64, irotavg_graph_time_kernel 200 + slot): shader-clock stamps of thread 0 of every
// workgroup of k_bcr_reduce at its phase boundaries, 16 slots per chunk
__device__ __forceinline__ void bcr_stamp(long long *st, int slot) {
    if (st && threadIdx.x == 0) st[(size_t)blockIdx.x * 16 + slot] = (long long)__builtin_readcyclecounter();
}

// NR: right-hand-side columns riding along (3: the three coordinates; 19: + 16 columns of the closures' incidence
// vectors, see bcr_solve)
template <int B, int NR = 3>
struct BcrDim {
    static constexpr int NC = 2 * B + NR;       // columns of W: P' | Q | R
    static constexpr int MT = (B + 15) / 16;    // 16-row tiles of a block
    static constexpr int NT = (NC + 15) / 16;   // 16-column tiles of W
    static constexpr int KS = B / 4;            // k-steps of a product over a block
    static constexpr int NT0 = B / 16;          // first column tile that holds a column of W_Q
};

// A-operand of the 16x16x4 product for row tile mt, k-step s: element (16 mt + lane % 16, 4 s + lane / 16) of the
// row-major B x B matrix M, or of its transpose; rows beyond B read as zero.
template <int B, bool TRANS>
__device__ __forceinline__ void bcr_load_a(const double *M, int lane, double (&aop)[BcrDim<B>::MT][BcrDim<B>::KS]) {
    const int li = lane & 15, lk = lane >> 4;
#pragma unroll
    for (int mt = 0; mt < BcrDim<B>::MT; mt++) {
        const int i = 16 * mt + li;
        const int ic = i < B ? i : B - 1;
#pragma unroll
        for (int s = 0; s < BcrDim<B>::KS; s++) {
            const int k = 4 * s + lk;
            const double v = TRANS ? M[k * B + ic] : M[ic * B + k];
            aop[mt][s] = i < B ? v : 0.0;
        }
    }
}

// out[mt][nt] += A (registers) x W (accumulator tiles of a previous product used as B-operands: register r of
// tile (mt, nt) holds the rows 16 mt + 4 r + lane / 16, i.e. k-step 4 mt + r)
template <int B, int NTA>
__device__ __forceinline__ void bcr_mul_w(const double (&aop)[BcrDim<B>::MT][BcrDim<B>::KS],
                                          const v4d (&w)[BcrDim<B>::MT][BcrDim<B>::NT],
                                          v4d (&out)[BcrDim<B>::MT][BcrDim<B>::NT]) {
    typedef BcrDim<B> Dm;
#pragma unroll
    for (int nt = NTA; nt < Dm::NT; nt++)
#pragma unroll
        for (int s = 0; s < Dm::KS; s++) {
            const double bop = w[s >> 2][nt][s & 3];
#pragma unroll
            for (int mt = 0; mt < Dm::MT; mt++)
                out[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[mt][s], bop, out[mt][nt], 0, 0, 0);
        }
}

// The matrix-core part of one block elimination. The 16-column tiles of W are independent of each other in all
// three products, so they are dealt to the waves of the elimination's group (one wave in the first round of a
// four-wave workgroup, two in the second, four in the third; four from the start in a sixteen-wave workgroup):
// this wave owns the tiles nt = part, part + WPE, ... and keeps them in w[][slot] between the phases (workgroup
// barriers lie in between). NTPW: slots of a wave.
template <int B, int NR, int NTPW>
struct BcrElim {
    typedef BcrDim<B, NR> Dm;
    v4d w[Dm::MT][NTPW];

    // W = Di^-1 [P' | Q | R] (Di^-1 in LDS), W -> global (the way back reads it), and the right neighbour:
    // D_c -= Q' W_Q, R_c -= Q' W_R
    // (hasQ = false: the block has no right neighbour -- the last block of the single-workgroup top, bcr_top_body; Wg =
    // nullptr: W is not wanted in memory)
    template <int WPE>
    __device__ __forceinline__ void phase1(int part, const double *Di, const double *P, bool hasP, const double *Q,
                                           const double *Ri, double *Dc, double *Rc, double *Wg, const double *zero,
                                           int lane, bool hasQ = true) {
        constexpr int NS = (Dm::NT + WPE - 1) / WPE;
        static_assert(NS <= NTPW, "slots");
        double aop[Dm::MT][Dm::KS];
        // (the inverse is symmetric: read by columns -- consecutive lanes, consecutive words -- instead of by rows, whose
        // stride of 2 B words puts sixteen lanes on four banks)
        bcr_load_a<B, true>(Di, lane, aop);
        const int lj = lane & 15, lk = lane >> 4;
        const double *base[NS];
        int stride[NS];
#pragma unroll
        for (int sl = 0; sl < NS; sl++) {
            const int nt = part + sl * WPE;
            const int j = 16 * nt + lj;
            base[sl] = zero;
            stride[sl] = 0;
            if (nt < Dm::NT) {
                if (j < B) {
                    if (hasP) {
                        base[sl] = P + j * B;  // column j of P' = row j of P
                        stride[sl] = 1;
                    }
                } else if (j < 2 * B) {
                    if (hasQ) {
                        base[sl] = Q + (j - B);
                        stride[sl] = B;
                    }
                } else if (j < Dm::NC) {
                    base[sl] = Ri + (j - 2 * B);
                    stride[sl] = NR;
                }
            }
#pragma unroll
            for (int mt = 0; mt < Dm::MT; mt++) w[mt][sl] = v4d{0.0, 0.0, 0.0, 0.0};
        }
        // k-steps outermost: the MT x NS accumulator tiles are independent chains, and a chain of dependent
        // f64 MFMAs issues one instruction per result latency, not per issue slot
#pragma unroll
        for (int s = 0; s < Dm::KS; s++) {
            double bop[NS];
#pragma unroll
            for (int sl = 0; sl < NS; sl++) bop[sl] = base[sl][(4 * s + lk) * stride[sl]];
#pragma unroll
            for (int sl = 0; sl < NS; sl++)
#pragma unroll
                for (int mt = 0; mt < Dm::MT; mt++)
                    w[mt][sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[mt][s], bop[sl], w[mt][sl], 0, 0, 0);
        }
        if (Wg) {
#pragma unroll
            for (int sl = 0; sl < NS; sl++) {
                const int j = 16 * (part + sl * WPE) + lj;
#pragma unroll
                for (int mt = 0; mt < Dm::MT; mt++)
#pragma unroll
                    for (int r = 0; r < 4; r++) {
                        const int row = 16 * mt + 4 * r + lk;
                        if (row < B && j < Dm::NC) Wg[row * Dm::NC + j] = w[mt][sl][r];
                    }
            }
        }
        if (!hasQ) return;
        bcr_load_a<B, true>(Q, lane, aop);
        v4d out[Dm::MT][NS];
#pragma unroll
        for (int sl = 0; sl < NS; sl++)
#pragma unroll
            for (int mt = 0; mt < Dm::MT; mt++) out[mt][sl] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < Dm::KS; s++)
#pragma unroll
            for (int sl = 0; sl < NS; sl++) {
                const int nt = part + sl * WPE;
                if (nt >= Dm::NT || nt < Dm::NT0) continue;
#pragma unroll
                for (int mt = 0; mt < Dm::MT; mt++)
                    out[mt][sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[mt][s], w[s >> 2][sl][s & 3], out[mt][sl], 0, 0, 0);
            }
#pragma unroll
        for (int sl = 0; sl < NS; sl++) {
            const int nt = part + sl * WPE;
            if (nt >= Dm::NT || nt < Dm::NT0) continue;
            const int c = 16 * nt + lj;
#pragma unroll
            for (int mt = 0; mt < Dm::MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = 16 * mt + 4 * r + lk;
                    if (row < B) {
                        if (c >= B && c < 2 * B)
                            Dc[row * B + c - B] -= out[mt][sl][r];
                        else if (c >= 2 * B && c < Dm::NC)
                            Rc[row * NR + c - 2 * B] -= out[mt][sl][r];
                    }
                }
        }
    }

    // the single-workgroup top keeps W in LDS for its way back, in the places its elimination has freed: W_P over
    // D_i^-1, W_Q over Q, W_R over R_i (after the barrier behind phase1: nobody reads those any more)
    template <int WPE>
    __device__ __forceinline__ void keep_w(int part, double *Di, double *Q, bool hasQ, double *Ri, int lane) const {
        constexpr int NS = (Dm::NT + WPE - 1) / WPE;
        const int lj = lane & 15, lk = lane >> 4;
#pragma unroll
        for (int sl = 0; sl < NS; sl++) {
            const int nt = part + sl * WPE;
            if (nt >= Dm::NT) continue;
            const int j = 16 * nt + lj;
#pragma unroll
            for (int mt = 0; mt < Dm::MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = 16 * mt + 4 * r + lk;
                    if (row < B) {
                        const double v = w[mt][sl][r];
                        if (j < B)
                            Di[row * B + j] = v;
                        else if (j < 2 * B) {
                            if (hasQ) Q[row * B + j - B] = v;
                        } else if (j < Dm::NC)
                            Ri[row * NR + j - 2 * B] = v;
                    }
                }
        }
    }

    // left neighbour, first half: the products P W (P is read by every wave of the group, and overwritten in
    // the second half -- a barrier lies in between when the group has more than one wave)
    template <int WPE>
    __device__ __forceinline__ void phase2_mul(int part, const double *P, v4d (&out)[Dm::MT][NTPW], int lane) {
        constexpr int NS = (Dm::NT + WPE - 1) / WPE;
        double aop[Dm::MT][Dm::KS];
        bcr_load_a<B, false>(P, lane, aop);
#pragma unroll
        for (int sl = 0; sl < NS; sl++)
#pragma unroll
            for (int mt = 0; mt < Dm::MT; mt++) out[mt][sl] = v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < Dm::KS; s++)
#pragma unroll
            for (int sl = 0; sl < NS; sl++) {
                if (part + sl * WPE >= Dm::NT) continue;
#pragma unroll
                for (int mt = 0; mt < Dm::MT; mt++)
                    out[mt][sl] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop[mt][s], w[s >> 2][sl][s & 3], out[mt][sl], 0, 0, 0);
            }
    }
    // second half: D_a -= P W_P (assigned when `assign`: the chunk's first contribution to the separator before
    // it), A[a, c] = -P W_Q written over P, R_a -= P W_R
    template <int WPE>
    __device__ __forceinline__ void phase2_put(int part, const v4d (&out)[Dm::MT][NTPW], double *P, double *Da,
                                               bool assign, double *Ra, int lane) {
        constexpr int NS = (Dm::NT + WPE - 1) / WPE;
        const int lj = lane & 15, lk = lane >> 4;
#pragma unroll
        for (int sl = 0; sl < NS; sl++) {
            const int nt = part + sl * WPE;
            if (nt >= Dm::NT) continue;
            const int c = 16 * nt + lj;
#pragma unroll
            for (int mt = 0; mt < Dm::MT; mt++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = 16 * mt + 4 * r + lk;
                    if (row < B) {
                        const double v = out[mt][sl][r];
                        if (c < B)
                            Da[row * B + c] = assign ? -v : Da[row * B + c] - v;
                        else if (c < 2 * B)
                            P[row * B + c - B] = -v;
                        else if (c < Dm::NC)
                            Ra[row * NR + c - 2 * B] -= v;
                    }
                }
        }
    }
};

// One row of the level-0 operator into the block arrays of a chunk: `row` lies in block lb (rows lb B ...), r = its
// row inside the block. Entries in block lb -> Dblk (with the diagonal), entries in block lb + 1 -> Gnext (the
// coupling lb -> lb + 1; nullptr: not wanted), entries in block lb - 1 -> GprevT, stored transposed (the coupling
// lb - 1 -> lb as the rows of block lb see it, by symmetry; nullptr: not wanted). The right-hand side -> Rblk.
// A thread owns its row (and, in GprevT, its column): plain read-modify-writes.
// Entries further away than the neighbouring blocks belong to long-range edges (loop closures): they are not part
// of the band operator this solver factorises -- their weight is taken back out of the diagonal (the entry is
// -w, the diagonal holds +w) and the closures re-enter by the Woodbury correction (bcr_solve). nfar > 0: the
// right-hand-side columns 3 .. 3 + nfar of the row get the incidence vectors of the closures of this pass.
template <int B, int NR, int NBT = 12>
__device__ __forceinline__ void bcr_gather_row(int row, int lb, int r, const int *__restrict__ sl_off,
                                               const int *__restrict__ col, const double *__restrict__ val,
                                               const double *__restrict__ diag, const double4 *__restrict__ rhs,
                                               double *blocks, int oD, int oGnext, int oGprevT, double *Rblk, int nfar,
                                               const int *__restrict__ far_i, const int *__restrict__ far_j) {
    // (Dblk, Gnext, GprevT = blocks + oD, + oGnext, + oGprevT; a negative offset: not wanted)
    double *Dblk = blocks + oD;
    const int sl = row >> 6, ln = row & 63;
    const int o0 = sl_off[sl], w = sl_off[sl + 1] - o0;
    const v2i *__restrict__ cp = reinterpret_cast<const v2i *>(col) + (size_t)(o0 / 2) * 64 + ln;
    const v2d *__restrict__ vp = reinterpret_cast<const v2d *>(val) + (size_t)(o0 / 2) * 64 + ln;
    const int c0 = lb * B;
    // (diagonal and right-hand side are requested ahead of the entries: behind them they were a memory round trip of
    // their own)
    const double dgv = diag[row];
    const double4 bb = rhs[row];
    // the row's entries in batches of NBT pairs: all loads of a batch are issued before the first value is
    // scattered (a load per scatter was a memory round trip per pair). NBT = 24: the raw blocks of a mixed level 1 inside
    // the single-launch upper reduction, where nine workgroups gather while sixty-four wait for them -- a row of the
    // headline graph (42 entries) is then ONE batch, one memory round trip less
    for (int q0 = 0; q0 < w / 2; q0 += NBT) {
        v2i cc[NBT];
        v2d vv[NBT];
#pragma unroll
        for (int u = 0; u < NBT; u++) {
            const int q = q0 + u < w / 2 ? q0 + u : w / 2 - 1;
            cc[u] = __builtin_nontemporal_load(&cp[(size_t)q * 64]);
            vv[u] = __builtin_nontemporal_load(&vp[(size_t)q * 64]);
        }
        // branch-free scatter by LDS adds without return (ds_add_f64: issued back to back, executed in program order, so
        // the sum of a row's duplicates keeps its order; a read-modify-write per entry was an LDS round trip and four
        // divergent branches each: a third of the load phase)
#pragma unroll
        for (int u = 0; u < NBT; u++) {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int c = h ? cc[u].y : cc[u].x;
                const double v = h ? vv[u].y : vv[u].x;
                const int gc = c - c0;
                // the target as an index: selects, no branches
                const bool inD = gc >= 0 && gc < B, inN = gc >= B && gc < 2 * B, inP = gc < 0 && gc >= -B;
                int idx = oD + r * B + r;  // long-range entry: its weight back out of the diagonal
                idx = inD ? oD + r * B + gc : idx;
                idx = inN ? oGnext + r * B + gc - B : idx;
                idx = inP ? oGprevT + (gc + B) * B + r : idx;
                const bool keep = q0 + u < w / 2 && v != 0.0 && !(inN && oGnext < 0) && !(inP && oGprevT < 0);  // (zero: padding)
                double *dst = blocks + idx;
                if (keep) atomicAdd(dst, v);
            }
        }
    }
    Dblk[r * B + r] += dgv;
    Rblk[r * NR + 0] = bb.x;
    Rblk[r * NR + 1] = bb.y;
    Rblk[r * NR + 2] = bb.z;
    if (NR > 3)
        for (int q = 0; q < nfar; q++) {
            const int fi = far_i[q], fj = far_j[q];
            if (fi == row || fj == row) Rblk[r * NR + 3 + q] = (fi == row ? 1.0 : 0.0) - (fj == row ? 1.0 : 0.0);
        }
}

// Where the k real blocks of a PARTIAL chunk sit among its eight positions. By default the first k (the rest is
// padding: identity blocks without couplings, whose eliminations are skipped) -- the chunk's separator, position 7, is
// then padding and its last real block is eliminated like any other: fine when nothing lies to its right. A SHARD's
// last block is coupled to the next rank and must survive as the separator at every level: its partial chunks place
// their blocks such that the last one sits at position 7 and every padding position is eliminated (= skipped) before
// it would be the neighbour of a real elimination in the (0 2 4 6)(1 5)(3) schedule. The coupling of the real block at
// position p to the next real block lives in slot p + 1, the one to the separator before the chunk in slot 0.
__constant__ signed char kBcrPlace[9][8] = {{-1, -1, -1, -1, -1, -1, -1, -1}, {7, -1, -1, -1, -1, -1, -1, -1},
                                             {3, 7, -1, -1, -1, -1, -1, -1},   {3, 5, 7, -1, -1, -1, -1, -1},
                                             {1, 3, 5, 7, -1, -1, -1, -1},     {1, 3, 5, 6, 7, -1, -1, -1},
                                             {1, 3, 4, 5, 6, 7, -1, -1},       {1, 2, 3, 4, 5, 6, 7, -1},
                                             {0, 1, 2, 3, 4, 5, 6, 7}};
__device__ __forceinline__ int bcr_pos(bool placed, int k, int t) { return placed ? (int)kBcrPlace[k][t] : t; }
__device__ __forceinline__ int bcr_ridx(bool placed, int k, int i) {
    if (!placed) return i < k ? i : -1;
    for (int t = 0; t < k; t++)
        if (kBcrPlace[k][t] == i) return t;
    return -1;
}

// One chunk of eight blocks per workgroup of NW waves (4, or 8 when the level has so few chunks that every
// workgroup has a CU to itself: the column tiles of W are then dealt to two waves in the first round and four afterwards).
// L0: the blocks are gathered from the level-0 SELL operator (off-diagonals), its diagonal and right-hand side;
// otherwise from the separator data of the level below (block j of this level = chunk j of that one):
//   D_j = sepD[j] + extD[j + 1],  R_j = sepR[j] + extR[j + 1],  G_j = extG[j + 1].
// Output per chunk c: W (seven blocks), sepD / sepR (block 7 after the eliminations), extD / extR (what the
// chunk's eliminations subtract from the separator of chunk c - 1), extG (coupling of that separator to block 7).
// TOP (one chunk, nothing before it): block 7 is solved and written to xtop.
// What one level writes for the next inside ONE launch (k_bcr_reduce_up) crosses the XCDs' L2s: device-scope atomic
// stores / loads (write-through, read at the coherence point) instead of fences -- a device-scope release writes back a
// whole L2, and tools/micro/xcdbar.hip measures a counter barrier + exchange among sixteen workgroups at 1.2 us this way.
template <bool COH>
__device__ __forceinline__ double bcr_ld(const double *p) {
    if (!COH) return *p;
    const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __longlong_as_double((long long)v);
}
template <bool COH>
__device__ __forceinline__ void bcr_st(double *p, double v) {
    if (!COH)
        *p = v;
    else
        __hip_atomic_store(reinterpret_cast<unsigned long long *>(p), (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
}

// The reduction of one chunk (the body of k_bcr_reduce; k_bcr_reduce_up runs it for the chunks of several levels in
// turn). sDG: 16 blocks, sR: 9 right-hand-side slots, sZ: 2 doubles of LDS. COH: the level below was written, and this
// level's separator data is read, inside the same launch (bcr_ld / bcr_st).
template <int B, int NR, bool L0, bool TOP, int NW, bool COH>
__device__ __forceinline__ void bcr_reduce_body(
    const int chunk, double (*sDG)[B * B], double (*sR)[B * NR], double *sZ,

    int nb, int nred, int n, const int *__restrict__ sl_off, const int *__restrict__ col, const double *__restrict__ val,
    const double *__restrict__ diag, const double4 *__restrict__ rhs, const double *__restrict__ inD,
    const double *__restrict__ inR, const double *__restrict__ inXD, const double *__restrict__ inXR,
    const double *__restrict__ inXG, double *__restrict__ W, double *__restrict__ sepD, double *__restrict__ sepR,
    double *__restrict__ extD, double *__restrict__ extR, double *__restrict__ extG, double *__restrict__ xtop, int dbg,
    int nfar, const int *__restrict__ far_i, const int *__restrict__ far_j, int ext0, const int *__restrict__ bptr,
    const int *__restrict__ bghost, const double *__restrict__ bval, const int *__restrict__ ghost_extcol, int place,
    long long *__restrict__ stamps, double *__restrict__ Dinvg, double *__restrict__ topDinv, int *__restrict__ deadctr,
    int reg) {
    typedef BcrDim<B, NR> Dm;
    constexpr int BB = B * B;
    double(*sD)[BB] = sDG;       // sD[0] becomes the chunk's contribution to the separator before it
    double(*sG)[BB] = sDG + 8;   // slot 0: coupling (separator before the chunk) -> block 0; slot j + 1: block j -> j + 1
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ext0: this handle is a SHARD of a sequence (dist.hip) and the separator before its first chunk is the last block
    // of the rank before it -- same algebra, the coupling comes from the boundary slots of the ghost views
    const bool hasExt = chunk > 0 || ext0;
    constexpr int NT_ = NW * 64;  // threads
    const int kreal = nb - chunk * 8 < 8 ? nb - chunk * 8 : 8;  // real blocks of this chunk
    const bool placed = place && kreal < 8;                      // ... placed so that the last one is the separator

    // ---- load ----
    // A chunk of an upper level whose blocks all come from the level below and sit at their own positions (every chunk
    // but a shard's partial ones and those of a mixed level 1 that hold raw level-0 blocks) is loaded by the FAST path
    // below: every word of LDS is written exactly once -- value, identity or zero -- so LDS is not cleared first (18
    // stores per thread and a barrier in front of the first load), the right-hand sides are requested together with the
    // blocks (they were a second memory round trip) and the index arithmetic is that of a contiguous copy (round 5: the
    // load phase of an upper level took 8 500 clocks, half of it address arithmetic of the general mapping).
    const bool fastload = !L0 && !placed && !(chunk * 8 + 8 > nred && nred < nb);
    if (!fastload) {
        for (int e = tid; e < 8 * BB; e += NT_) {
            (&sD[0][0])[e] = 0.0;
            (&sG[0][0])[e] = 0.0;
        }
        for (int e = tid; e < 9 * B * NR; e += NT_) (&sR[0][0])[e] = 0.0;
        if (tid < 2) sZ[tid] = 0.0;
    }
    bcr_stamp(stamps, 0);
    if (!fastload) __syncthreads();
    bcr_stamp(stamps, 1);
    if (!L0 && fastload) {
        // block gb = chunk 8 + i of this level = chunk gb of the level below; (gb BB + el) = chunk 8 BB + e: contiguous
        constexpr int NIT = (8 * BB + NT_ - 1) / NT_, U = NIT <= 12 ? NIT : 8;
        constexpr int NIR = (8 * B * NR + NT_ - 1) / NT_;
        const size_t cb = (size_t)chunk * 8 * BB, cr = (size_t)chunk * 8 * B * NR;
        double vr[NIR], vy[NIR];
#pragma unroll
        for (int u = 0; u < NIR; u++) {
            const int e = tid + NT_ * u, i = e / (B * NR);
            const bool ok = e < 8 * B * NR && i < kreal;
            vr[u] = ok ? bcr_ld<COH>(inR + cr + e) : 0.0;
            vy[u] = ok && chunk * 8 + i + 1 < nred ? bcr_ld<COH>(inXR + cr + B * NR + e) : 0.0;
        }
        for (int u0 = 0; u0 < NIT; u0 += U) {
            double vd[U], vx[U], vg[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int e = tid + NT_ * (u0 + u), i = e / BB;
                const bool ok = e < 8 * BB && i < kreal;
                vd[u] = ok ? bcr_ld<COH>(inD + cb + e) : 0.0;
                vx[u] = ok && chunk * 8 + i + 1 < nred ? bcr_ld<COH>(inXD + cb + BB + e) : 0.0;
                vg[u] = ok && (chunk * 8 + i > 0 || ext0) ? bcr_ld<COH>(inXG + cb + e) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int e = tid + NT_ * (u0 + u), i = e / BB, el = e - i * BB;
                if (e >= 8 * BB) continue;
                // (a position without a block: the identity, its eliminations are skipped)
                (&sD[0][0])[e] = i < kreal ? vd[u] + vx[u] : (el % (B + 1) == 0 ? 1.0 : 0.0);
                (&sG[0][0])[e] = vg[u];
            }
        }
#pragma unroll
        for (int u = 0; u < NIR; u++) {
            const int e = tid + NT_ * u;
            if (e < 8 * B * NR) (&sR[1][0])[e] = vr[u] + vy[u];
        }
        for (int e = tid; e < B * NR; e += NT_) sR[0][e] = 0.0;
        if (tid < 2) sZ[tid] = 0.0;
    } else if (L0) {
        const int row0 = chunk * 8 * B;
        for (int t = tid; t < kreal * B; t += NT_) {
            const int row = row0 + t, blk = t / B, r = t - blk * B;
            const int ps = bcr_pos(placed, kreal, blk);
            if (row < n)
                bcr_gather_row<B, NR>(row, chunk * 8 + blk, r, sl_off, col, val, diag, rhs, &sDG[0][0], ps * BB,
                                      ps < 7 ? (8 + ps + 1) * BB : -1, blk == 0 ? 8 * BB : -1, sR[ps + 1], nfar, far_i,
                                      far_j);
            else
                sD[ps][r * B + r] = 1.0;
        }
        // (position 0 of a placed chunk stays zero: sD[0] collects the chunk's contribution to the separator before it,
        // and nothing assigns it first when no real block sits there)
        for (int i = placed ? 1 : 0; i < 8; i++)
            if (bcr_ridx(placed, kreal, i) < 0)
                for (int e = tid; e < B; e += NT_) sD[i][e * B + e] = 1.0;
        if (ext0 && chunk == 0) {
            // coupling (last block of the previous rank) -> block 0: the ghost entries of block 0's rows, -w each
            // (k_assemble0w leaves the weight of every boundary slot in bval); a thread owns column r of sG[0]
            for (int r = tid; r < B && r < n; r += NT_)
                for (int sl = bptr[r]; sl < bptr[r + 1]; sl++) {
                    const int gi = bghost[sl];
                    const int ec = gi >= 0 ? ghost_extcol[gi] : -1;
                    if (ec >= 0) sG[0][ec * B + r] -= bval[sl];
                }
        }
    } else {
        // blocks below nred come from the level below; the others (a mixed level 1, see bcr_alloc) are blocks of
        // the level-0 operator that no chunk reduced: block gb is level-0 block 8 nred + (gb - nred). Every block
        // brings the coupling from its predecessor along (slot i; slot 0: from the separator before the chunk).
        // every load of the launch is requested before the first value is used (a loop over the eight positions with
        // its loads inside was a memory round trip per position: 5 of the 8 us a launch costs before its first sweep)
        auto where = [&](int i, int &gb, int &slot) {
            const int t = bcr_ridx(placed, kreal, i);
            gb = t >= 0 ? chunk * 8 + t : -1;
            slot = t > 0 ? bcr_pos(placed, kreal, t - 1) + 1 : 0;
            return t;
        };
        constexpr int NIT = (8 * BB + NT_ - 1) / NT_, U = NIT <= 12 ? NIT : 8;
        // (a chunk of a mixed level that holds raw level-0 blocks only has nothing to fetch from the level below: the
        // positions without a block become the identity and the walk over the eight positions is skipped -- nine such
        // workgroups are what the sixty-four others of level 1 wait for at 100k views)
        const bool allraw = !placed && chunk * 8 >= nred;
        if (allraw)
            for (int e = tid; e < (8 - kreal) * B; e += NT_) sD[kreal + e / B][(e % B) * (B + 1)] = 1.0;
        for (int u0 = 0; u0 < NIT && !allraw; u0 += U) {
            double vd[U], vx[U], vg[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int e = tid + NT_ * (u0 + u), i = e / BB, el = e - i * BB;
                int gb, slot;
                const int t = e < 8 * BB ? where(i, gb, slot) : -1;
                const bool ok = t >= 0 && gb < nred;
                vd[u] = ok ? bcr_ld<COH>(inD + (size_t)gb * BB + el) : 0.0;
                vx[u] = ok && gb + 1 < nred ? bcr_ld<COH>(inXD + (size_t)(gb + 1) * BB + el) : 0.0;
                vg[u] = ok && (gb > 0 || ext0) ? bcr_ld<COH>(inXG + (size_t)gb * BB + el) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int e = tid + NT_ * (u0 + u), i = e / BB, el = e - i * BB;
                if (e >= 8 * BB) continue;
                int gb, slot;
                const int t = where(i, gb, slot);
                if (t >= 0 && gb < nred) {
                    sD[i][el] = vd[u] + vx[u];
                    if (gb > 0 || ext0) sG[slot][el] = vg[u];
                } else if (t < 0 && !(placed && i == 0) && el / B == el % B) {  // (sD[0] of a placed chunk: see the level-0 branch)
                    sD[i][el] = 1.0;
                }
            }
        }
        if (!allraw) {
            constexpr int NIR = (8 * B * NR + NT_ - 1) / NT_;
            double vr[NIR], vy[NIR];
#pragma unroll
            for (int u = 0; u < NIR; u++) {
                const int e = tid + NT_ * u, i = e / (B * NR), el = e - i * (B * NR);
                int gb, slot;
                const int t = e < 8 * B * NR ? where(i, gb, slot) : -1;
                const bool ok = t >= 0 && gb < nred;
                vr[u] = ok ? bcr_ld<COH>(inR + (size_t)gb * B * NR + el) : 0.0;
                vy[u] = ok && gb + 1 < nred ? bcr_ld<COH>(inXR + (size_t)(gb + 1) * B * NR + el) : 0.0;
                if (ok) sR[i + 1][el] = vr[u] + vy[u];
            }
        }
        if (chunk * 8 + 8 > nred && nred < nb) {
            for (int t = tid; t < 8 * B; t += NT_) {
                const int i = t / B, r = t - i * B, gb = chunk * 8 + i;
                if (gb >= nred && gb < nb) {
                    const int lb = 8 * nred + (gb - nred), row = lb * B + r;
                    if (row < n)
                        bcr_gather_row<B, NR, (NW == 8 && B <= 24 ? 24 : 12)>(row, lb, r, sl_off, col, val, diag, rhs, &sDG[0][0],
                                                                            i * BB, -1, gb > 0 ? (8 + i) * BB : -1, sR[i + 1],
                                                                            nfar, far_i, far_j);
                    else
                        sD[i][r * B + r] = 1.0;
                }
            }
        }
    }
    __syncthreads();
    bcr_stamp(stamps, 2);

    // ---- three rounds of eliminations: (0 2 4 6) (1 5) (3) ----
    constexpr int WPE0 = NW / 4 < Dm::NT ? NW / 4 : Dm::NT;  // waves per elimination in the first round
    constexpr int NTPW = (Dm::NT + WPE0 - 1) / WPE0;
    BcrElim<B, NR, NTPW> E;
    v4d out[Dm::MT][NTPW];
#define IRH_BCR_ROUND(RND)                                                                                          \
    {                                                                                                               \
        constexpr int NE = 4 >> RND;                                                                                \
        constexpr int WPE = NW / NE < Dm::NT ? NW / NE : Dm::NT;                                                    \
        /* (two waves per elimination, four eliminations: the wave that owns the column tiles 1 and 3 issues 72 products, */ \
        /* the other one 60, and wave w sits on SIMD w % 4 -- the roles are swapped in eliminations 2 and 3 so that every */ \
        /* SIMD gets one wave of each kind: 132 instead of 144 / 120 products of 64 clocks per SIMD)                      */ \
        const int e = wave / WPE, part0 = wave - e * WPE;                                                           \
        const int part = (WPE == 2 && NE == 4) ? (part0 ^ ((e >> 1) & 1)) : part0;                                  \
        int i = -1, a = -1, c = 7;                                                                                  \
        if (e < NE) {                                                                                               \
            if (RND == 0) {                                                                                         \
                i = 2 * e;                                                                                          \
                a = i - 1;                                                                                          \
                c = i + 1;                                                                                          \
            } else if (RND == 1) {                                                                                  \
                i = 1 + 4 * e;                                                                                      \
                a = e == 0 ? -1 : 3;                                                                                \
                c = e == 0 ? 3 : 7;                                                                                 \
            } else {                                                                                                \
                i = 3;                                                                                              \
            }                                                                                                       \
        }                                                                                                           \
        const bool active = i >= 0 && bcr_ridx(placed, kreal, i) >= 0;                                              \
        const bool hasP = a >= 0 || hasExt;                                                                         \
        /* the sweep of elimination e runs on wave e: the first four waves of a workgroup sit on four different */   \
        /* SIMDs (waves e WPE of an eight-wave workgroup share two, and two sweeps on one SIMD take 7 us, not 4) */  \
        const int isw = wave < NE ? (RND == 0 ? 2 * wave : RND == 1 ? 1 + 4 * wave : 3) : -1;                       \
        if (isw >= 0 && bcr_ridx(placed, kreal, isw) >= 0 && !(dbg & 1))                                            \
            bcr_invert<B>(sD[isw], lane, deadctr, reg != 0);                                                        \
        __syncthreads();                                                                                            \
        /* closures: the inverse of every eliminated block is kept (bcr_closures) */                                \
        if (Dinvg && active && part == 0)                                                                           \
            for (int o = lane; o < BB; o += 64) Dinvg[((size_t)chunk * 7 + i) * BB + o] = sD[i][o];                 \
        bcr_stamp(stamps, 3 + 4 * RND);                                                                             \
        if (active && !(dbg & 2))                                                                                   \
            E.template phase1<WPE>(part, sD[i], sG[a + 1], hasP, sG[i + 1], sR[i + 1], sD[c], sR[c + 1],            \
                                   W + ((size_t)chunk * 7 + i) * B * Dm::NC, sZ, lane);                             \
        __syncthreads();                                                                                            \
        bcr_stamp(stamps, 4 + 4 * RND);                                                                             \
        if (active && hasP && !(dbg & 4)) E.template phase2_mul<WPE>(part, sG[a + 1], out, lane);                   \
        if (WPE > 1) __syncthreads();                                                                               \
        bcr_stamp(stamps, 5 + 4 * RND);                                                                             \
        if (active && hasP && !(dbg & 4))                                                                           \
            E.template phase2_put<WPE>(part, out, sG[a + 1], a >= 0 ? sD[a] : sD[0], a < 0 && RND == 0, sR[a + 1],  \
                                       lane);                                                                       \
        __syncthreads();                                                                                            \
        bcr_stamp(stamps, 6 + 4 * RND);                                                                             \
    }
    IRH_BCR_ROUND(0)
    IRH_BCR_ROUND(1)
    IRH_BCR_ROUND(2)
#undef IRH_BCR_ROUND

    // ---- store ----
    if (TOP) {
        if (wave == 0) {
            if (bcr_ridx(placed, kreal, 7) >= 0) {
                bcr_invert<B>(sD[7], lane, deadctr, reg != 0);
                if (topDinv)
                    for (int o = lane; o < BB; o += 64) topDinv[o] = sD[7][o];
                for (int o = lane; o < B * NR; o += 64) {
                    const int k = o / NR, q = o - NR * k;
                    double s = 0.0;
                    for (int cidx = 0; cidx < B; cidx++) s += sD[7][k * B + cidx] * sR[8][cidx * NR + q];
                    xtop[o] = s;
                }
            } else {
                for (int o = lane; o < B * NR; o += 64) xtop[o] = 0.0;
            }
        }
    } else {
        for (int e = tid; e < BB; e += NT_) {
            bcr_st<COH>(sepD + (size_t)chunk * BB + e, sD[7][e]);
            if (hasExt) {
                bcr_st<COH>(extD + (size_t)chunk * BB + e, sD[0][e]);
                bcr_st<COH>(extG + (size_t)chunk * BB + e, sG[0][e]);
            }
        }
        for (int e = tid; e < B * NR; e += NT_) {
            bcr_st<COH>(sepR + (size_t)chunk * B * NR + e, sR[8][e]);
            if (hasExt) bcr_st<COH>(extR + (size_t)chunk * B * NR + e, sR[0][e]);
        }
    }
    bcr_stamp(stamps, 15);
}

template <int B, int NR, bool L0, bool TOP, int NW>
__global__ __launch_bounds__(NW * 64, (NW == 4 && B <= 24 && NR == 3 ? 2 : 1)) void k_bcr_reduce(

    int nb, int nred, int n, const int *__restrict__ sl_off, const int *__restrict__ col, const double *__restrict__ val,
    const double *__restrict__ diag, const double4 *__restrict__ rhs, const double *__restrict__ inD,
    const double *__restrict__ inR, const double *__restrict__ inXD, const double *__restrict__ inXR,
    const double *__restrict__ inXG, double *__restrict__ W, double *__restrict__ sepD, double *__restrict__ sepR,
    double *__restrict__ extD, double *__restrict__ extR, double *__restrict__ extG, double *__restrict__ xtop, int dbg,
    int nfar, const int *__restrict__ far_i, const int *__restrict__ far_j, int ext0, const int *__restrict__ bptr,
    const int *__restrict__ bghost, const double *__restrict__ bval, const int *__restrict__ ghost_extcol, int place,
    long long *__restrict__ stamps, double *__restrict__ Dinvg, double *__restrict__ topDinv, int *__restrict__ deadctr,
    int reg) {
    __shared__ double sDG[16][B * B];
    __shared__ double sR[9][B * NR];  // slot 0: contribution to the right-hand side of the separator before the chunk
    __shared__ double sZ[2];
    bcr_reduce_body<B, NR, L0, TOP, NW, false>(blockIdx.x, sDG, sR, sZ, nb, nred, n, sl_off, col, val, diag, rhs, inD, inR, inXD, inXR,
                                               inXG, W, sepD, sepR, extD, extR, extG, xtop, dbg, nfar, far_i, far_j, ext0, bptr,
                                               bghost, bval, ghost_extcol, place, stamps, Dinvg, topDinv, deadctr, reg);
}

// ---------------------------------------------------------------------------------------------
// The top of the hierarchy in ONE workgroup: up to sixteen blocks (round 5; until then a level of 9 ... 64 blocks was
// reduced by chunks of eight to a top of <= 8 -- at 100k views the last two levels held 10 and 2 blocks and cost a load,
// four rounds, a store and a level boundary more than the 10 blocks need). The blocks of the top level are loaded once;
// every round eliminates up to four blocks no two of which are neighbours (four sweeps on four SIMDs, their products
// dealt to the eight waves as in a chunk), by a schedule the host writes (bcr_top_schedule: the first four even positions
// of the list of blocks still alive -- 10 blocks: 10 -> 6 -> 3 -> 1 in three rounds, 16: five); the block that is left
// is solved. W stays in LDS, in the places an elimination frees (W_P over D_i^-1, W_Q over the coupling to the right
// neighbour, W_R over R_i), so the way back of the top level follows at once: the workgroup writes the solutions of ALL
// its blocks, and the ways back of the levels below start from them (k_bcr_back_top / k_bcr_back: one level less).
// The coupling of block x to the next block alive lives in sG[x]; a fill-in A[a, c] takes the place of A[a, i].
// Three right-hand sides, B <= 24 (LDS: (2 n - 1) blocks + 2 n right-hand-side slots = 158 KB for n = 16, B = 24), no
// closures (their step programs know chunks only), not a shard.
// ---------------------------------------------------------------------------------------------
static BcrTopSched bcr_top_schedule(int n) {
    BcrTopSched T;
    memset(&T, 0, sizeof(T));
    T.n = n;
    std::vector<int> alive(n);
    for (int k = 0; k < n; k++) alive[k] = k;
    while (alive.size() > 1 && T.nr < kTopRounds) {
        const int r = T.nr++;
        std::vector<int> next;
        int ne = 0;
        for (size_t p = 0; p < alive.size(); p++) {
            if (p % 2 == 0 && ne < 4) {
                T.i[r][ne] = (signed char)alive[p];
                T.a[r][ne] = (signed char)(p > 0 ? alive[p - 1] : -1);
                T.c[r][ne] = (signed char)(p + 1 < alive.size() ? alive[p + 1] : -1);
                ne++;
            } else {
                next.push_back(alive[p]);
            }
        }
        T.ne[r] = (signed char)ne;
        alive.swap(next);
    }
    T.last = alive[0];
    return T;
}
static size_t bcr_top_lds(int B, int n) {
    return ((size_t)(2 * n - 1) * B * B + (size_t)2 * n * B * 3 + 2) * sizeof(double) + sizeof(BcrTopSched);
}

template <int B, int NW, bool COH>
__device__ __forceinline__ void bcr_top_body(double *smem, const BcrTopSched *__restrict__ Tg, const double *__restrict__ inD,
                                             const double *__restrict__ inR, const double *__restrict__ inXD,
                                             const double *__restrict__ inXR, const double *__restrict__ inXG,
                                             double *__restrict__ xout, int dbg, long long *__restrict__ stamps) {
    constexpr int NR = 3;
    typedef BcrDim<B, NR> Dm;
    constexpr int BB = B * B, XB = B * NR, NT_ = NW * 64;
    static_assert(NW == 8, "eight waves");
    const int n = Tg->n;
    double(*sD)[BB] = reinterpret_cast<double(*)[BB]>(smem);
    double(*sG)[BB] = sD + n;  // n - 1 slots
    double(*sR)[XB] = reinterpret_cast<double(*)[XB]>(smem + (size_t)(2 * n - 1) * BB);
    double(*sX)[XB] = sR + n;
    double *sZ = &sX[n][0];
    // (the schedule is read with indices that are not known at compile time, by every wave, in every round: LDS)
    BcrTopSched &T = *reinterpret_cast<BcrTopSched *>(sZ + 2);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    static_assert(sizeof(BcrTopSched) % 4 == 0, "copied by words");
    if (tid < (int)(sizeof(BcrTopSched) / 4)) reinterpret_cast<int *>(&T)[tid] = reinterpret_cast<const int *>(Tg)[tid];
    bcr_stamp(stamps, 0);
    // ---- load: D_j = sepD[j] + extD[j + 1], R_j = sepR[j] + extR[j + 1], coupling j - 1 -> j = extG[j] ----
    {
        constexpr int U = 6;
        const int nD = n * BB, nG = (n - 1) * BB, nR = n * XB;
        for (int e0 = 0; e0 < nD; e0 += U * NT_) {
            double vd[U], vx[U], vg[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int e = e0 + tid + NT_ * u;
                vd[u] = e < nD ? bcr_ld<COH>(inD + e) : 0.0;
                vx[u] = e < nG ? bcr_ld<COH>(inXD + BB + e) : 0.0;
                vg[u] = e < nG ? bcr_ld<COH>(inXG + BB + e) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int e = e0 + tid + NT_ * u;
                if (e < nD) (&sD[0][0])[e] = vd[u] + vx[u];
                if (e < nG) (&sG[0][0])[e] = vg[u];
            }
        }
        for (int e = tid; e < nR; e += NT_) {
            const double vr = bcr_ld<COH>(inR + e);
            const double vy = e < nR - XB ? bcr_ld<COH>(inXR + XB + e) : 0.0;
            (&sR[0][0])[e] = vr + vy;
        }
        if (tid < 2) sZ[tid] = 0.0;
    }
    __syncthreads();
    bcr_stamp(stamps, 1);
    // ---- the rounds ----
    for (int r = 0; r < T.nr; r++) {
        const int ne = T.ne[r];
        if (wave < ne && !(dbg & 1)) bcr_invert<B>(sD[T.i[r][wave]], lane);
        __syncthreads();
        bcr_stamp(stamps, 2 + 2 * r);
        auto products = [&](auto wpe_tag) {
            constexpr int WPE = decltype(wpe_tag)::value;
            const int e = wave / WPE, part0 = wave - e * WPE;
            const int part = WPE == 2 ? (part0 ^ ((e >> 1) & 1)) : part0;
            const bool active = e < ne;
            const int i = active ? T.i[r][e] : 0, a = active ? T.a[r][e] : -1, c = active ? T.c[r][e] : -1;
            const bool hasP = a >= 0, hasQ = c >= 0;
            BcrElim<B, NR, 2> E;
            v4d out[Dm::MT][2];
            if (active && !(dbg & 2))
                E.template phase1<WPE>(part, sD[i], hasP ? sG[a] : sZ, hasP, hasQ ? sG[i] : sZ, sR[i], hasQ ? sD[c] : sD[i],
                                       hasQ ? sR[c] : sR[i], (double *)nullptr, sZ, lane, hasQ);
            __syncthreads();
            if (active && hasP && !(dbg & 4)) E.template phase2_mul<WPE>(part, sG[a], out, lane);
            if (active && !(dbg & 2)) E.template keep_w<WPE>(part, sD[i], hasQ ? sG[i] : sZ, hasQ, sR[i], lane);
            __syncthreads();
            if (active && hasP && !(dbg & 4)) E.template phase2_put<WPE>(part, out, sG[a], sD[a], false, sR[a], lane);
            __syncthreads();
        };
        if (ne > 2) products(std::integral_constant<int, 2>());
        else products(std::integral_constant<int, 4>());
        bcr_stamp(stamps, 3 + 2 * r);
    }
    // ---- the block that is left ----
    const int last = T.last;
    if (wave == 0) {
        bcr_invert<B>(sD[last], lane);
        for (int o = lane; o < XB; o += 64) {
            const int k = o / NR, q = o - NR * k;
            double acc = 0.0;
            for (int cidx = 0; cidx < B; cidx++) acc += sD[last][k * B + cidx] * sR[last][cidx * NR + q];
            sX[last][o] = acc;
        }
    }
    __syncthreads();
    bcr_stamp(stamps, 14);
    // ---- the way back: x_i = W_R - W_P x_a - W_Q x_c, rounds in reverse; a unit of work = four rows of a block ----
    {
        constexpr int UPB = B / 4, NCT = (2 * B + 15) / 16;
        const int lk = lane >> 4, lp = lane & 15;
        for (int r = T.nr - 1; r >= 0; r--) {
            const int ne = T.ne[r];
            for (int u = wave; u < ne * UPB; u += NW) {
                const int e = u / UPB, it = u - e * UPB;
                const int i = T.i[r][e], a = T.a[r][e], c = T.c[r][e];
                const double *xa = sX[a >= 0 ? a : 0], *xc = sX[c >= 0 ? c : 0];
                const int k = 4 * it + lk;
                double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int t = 0; t < NCT; t++) {
                    const int cidx = lp + 16 * t;
                    if (cidx < B) {
                        if (a >= 0) {
                            const double wv = sD[i][k * B + cidx];
                            s0 += wv * xa[cidx * 3 + 0];
                            s1 += wv * xa[cidx * 3 + 1];
                            s2 += wv * xa[cidx * 3 + 2];
                        }
                    } else if (cidx < 2 * B) {
                        if (c >= 0) {
                            const double wv = sG[i][k * B + cidx - B];
                            s0 += wv * xc[(cidx - B) * 3 + 0];
                            s1 += wv * xc[(cidx - B) * 3 + 1];
                            s2 += wv * xc[(cidx - B) * 3 + 2];
                        }
                    }
                }
                s0 = bcr_row16_sum(s0);
                s1 = bcr_row16_sum(s1);
                s2 = bcr_row16_sum(s2);
                if (lp == 0) {
                    sX[i][k * 3 + 0] = sR[i][k * 3 + 0] - s0;
                    sX[i][k * 3 + 1] = sR[i][k * 3 + 1] - s1;
                    sX[i][k * 3 + 2] = sR[i][k * 3 + 2] - s2;
                }
            }
            __syncthreads();
        }
    }
    for (int e = tid; e < n * XB; e += NT_) xout[e] = (&sX[0][0])[e];
    bcr_stamp(stamps, 15);
}

// the top alone (a launch of its own: timing of single levels, a handle whose single-launch reduction was not admitted)
template <int B>
__global__ __launch_bounds__(512, 1) void k_bcr_top(const BcrTopSched *__restrict__ T, const double *__restrict__ inD, const double *__restrict__ inR,
                                                    const double *__restrict__ inXD, const double *__restrict__ inXR,
                                                    const double *__restrict__ inXG, double *__restrict__ xout, int dbg,
                                                    long long *__restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) double top_smem[];
    bcr_top_body<B, 8, false>(top_smem, T, inD, inR, inXD, inXR, inXG, xout, dbg, stamps);
}

// The reductions of ALL levels above level 0 in ONE launch (round 4). Those levels hold 1.6 % of the rows; each was a
// launch of 73 / 10 / 2 / 1 workgroups that spends ~5 us of its 20 - 38 us starting, filling and draining. Here workgroup c
// runs chunk c of every level that has one, level after level; between two levels the workgroups that go on wait for a
// counter every workgroup of the level before has passed. What the first attempt at this lost (round 3: 325 instead of
// 290 us per solve) was the way the data crossed the level boundary -- a device-scope release and acquire per workgroup
// and boundary, i.e. L2 write-backs and invalidates, 7 us each. Now the separator data is WRITTEN by device-scope
// atomic stores and READ by device-scope atomic loads (bcr_st / bcr_ld: the coherence point of the eight L2s, no
// fence), the counter is a relaxed atomic: ~1.2 us per boundary (tools/micro/xcdbar.hip, sixteen workgroups on eight
// XCDs). All workgroups of the launch fit the chip at once (one per CU), so nobody waits for a workgroup that cannot
// start. A wait that does not end (~1 s) poisons the top separator: the solve then fails with IROTAVG_ERR_SOLVER at the
// next score instead of hanging the device. W goes to memory as before: the ways back are later launches.
// The bodies of the levels are separate FUNCTIONS (round 5): inlined into one kernel the three of them -- a chunk, a chunk
// that is the top, the sixteen-block top -- shared one register allocation with the level loop around them, and the
// kernel spilled (36 registers in round 4, 155 with the sixteen-block top) where each body alone needs 124 - 212 and
// spills nothing. A call per level costs nothing against 30 us of work. Their arguments live in device memory (written
// once per handle: only the generation, the debug word and the stamp array change from solve to solve and travel as
// kernel arguments); LDS is the launch's dynamic allocation, named by the functions themselves.
// (16-byte alignment: the sweeps read LDS by ds_read_b128; behind the kernel's one static word an 8-byte-aligned base made
// every such read a split one and a sweep 3.5 x slower)
extern __shared__ __attribute__((aligned(16))) double up_smem[];
typedef IRH_GPTR(const BcrUpArgs) gp_args;
template <int B, bool TOP>
__device__ __noinline__ void bcr_up_chunk(gp_args A, int i, int dbg, long long *stl) {
    constexpr int NR = 3;
    IRH_GPTR(const BcrUpLevel) L = &A->lev[i];
    double(*sDG)[B * B] = reinterpret_cast<double(*)[B * B]>(up_smem);
    double(*sR)[B * NR] = reinterpret_cast<double(*)[B * NR]>(up_smem + 16 * B * B);
    double *sZ = up_smem + 16 * B * B + 9 * B * NR;
    bcr_reduce_body<B, NR, false, TOP, 8, true>(
        blockIdx.x, sDG, sR, sZ, L->nb, L->nred, A->n, (const int *)A->sl_off, (const int *)A->col, (const double *)A->val,
        (const double *)A->diag, (const double4 *)A->rhs, (const double *)L->inD, (const double *)L->inR, (const double *)L->inXD,
        (const double *)L->inXR, (const double *)L->inXG, (double *)L->W, (double *)L->sepD, (double *)L->sepR, (double *)L->extD,
        (double *)L->extR, (double *)L->extG, (double *)A->xtop, dbg, 0, (const int *)nullptr, (const int *)nullptr, 0, (const int *)nullptr, (const int *)nullptr,
        (const double *)nullptr, (const int *)nullptr, 0, stl ? stl - (size_t)blockIdx.x * 16 : nullptr, (double *)nullptr,
        (double *)nullptr, (int *)nullptr, 0);   // (bcr_stamp adds 16 x the workgroup's number)
}
template <int B>
__device__ __noinline__ void bcr_up_top(gp_args A, int i, int dbg, long long *stl) {
    IRH_GPTR(const BcrUpLevel) L = &A->lev[i];
    bcr_top_body<B, 8, true>(up_smem, (const BcrTopSched *)&A->top, (const double *)L->inD, (const double *)L->inR,
                             (const double *)L->inXD, (const double *)L->inXR, (const double *)L->inXG, (double *)A->xtop_all, dbg,
                             stl ? stl - (size_t)blockIdx.x * 16 : nullptr);
}
template <int B>
__global__ __launch_bounds__(512, 1) void k_bcr_reduce_up(const BcrUpArgs *__restrict__ Ap, unsigned gen, int dbg,
                                                          long long *__restrict__ stamps) {
    constexpr int NR = 3;
    __shared__ int s_ok;
    const gp_args Ag = (gp_args)Ap;
    const BcrUpArgs &A = *Ap;
    const int wg = blockIdx.x;
    const int nl = A.nl;
    bool fine = true;
    for (int i = 0; i < nl; i++) {
        if (wg >= A.lev[i].nch) break;  // (the chunk counts shrink level by level)
        long long *stl = stamps && wg == (dbg >> 8) ? stamps + 32 * i : nullptr;  // (the workgroup that stamps: dbg >> 8)
        if (stl && threadIdx.x == 0) stl[16] = (long long)__builtin_readcyclecounter();
        // (slots 20 ...: where the workgroup's waves sit -- HW_ID: wave [3:0], SIMD [5:4], CU [11:8], SE [15:13])
        if (stl && (threadIdx.x & 63) == 0) stl[20 + (threadIdx.x >> 6)] = 0x10000 + (long long)__builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));
        if (i > 0) {
            if (threadIdx.x == 0) {
                const unsigned target = gen * (unsigned)A.lev[i - 1].nch;
                int good = 0;
                for (int spin = 0; spin < (1 << 21); spin++) {
                    // (a difference: the counter may wrap)
                    if ((int)(__hip_atomic_load((const unsigned *)A.cnt + (i - 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) {
                        good = 1;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(2);
                }
                if (!good) __hip_atomic_store((int *)A.fail, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                s_ok = good;
            }
            __syncthreads();
            fine = fine && s_ok != 0;
        }
        if (stl && threadIdx.x == 0) stl[17] = (long long)__builtin_readcyclecounter();
        if (i == nl - 1) {
            // (the top workgroup has waited for every workgroup of the level below, and a workgroup sets the word before it
            // counts itself)
            const bool bad = !fine || __hip_atomic_load((const int *)A.fail, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
            // bad: NaN into every entry -- it then reaches every solution row on the ways back, no view takes a step
            // (step_apply leaves a rotation alone when its step is not finite) and the score is NaN
            if (A.top16) {
                if (!bad) bcr_up_top<B>(Ag, i, dbg, stl);
                else
                    for (int e = threadIdx.x; e < A.top.n * B * NR; e += 512) A.xtop_all[e] = __builtin_nan("");
            } else {
                if (!bad) bcr_up_chunk<B, true>(Ag, i, dbg, stl);
                else
                    for (int e = threadIdx.x; e < B * NR; e += 512) A.xtop[e] = __builtin_nan("");
            }
        } else {
            if (fine) bcr_up_chunk<B, false>(Ag, i, dbg, stl);
            if (stl && threadIdx.x == 0) stl[18] = (long long)__builtin_readcyclecounter();
            // every thread's separator stores have been acknowledged before the workgroup is counted
            __builtin_amdgcn_s_waitcnt(0);
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_fetch_add((unsigned *)A.cnt + i, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (stl && threadIdx.x == 0) stl[19] = (long long)__builtin_readcyclecounter();
        }
    }
}

// k_bcr_reduce_up's workgroups WAIT for each other, so all workgroups of a launch must be resident at the same time -- and
// so must those of every other such launch that is dispatched next to it (l1ra's three chains, other handles, other
// sessions): two launches that each hold a part of the device and spin for the rest would wait until the bounded spin
// gives up. The launches of one process therefore RESERVE their share of the device first: a launch of N workgroups
// costs N / (workgroups of this kernel that fit the device, from the occupancy query) and is admitted while the sum over
// all handles stays within the device; a handle keeps its share (its launches are ordered on its stream: the largest
// counts, not the sum) until it is idle. A launch that is not admitted runs level by level, as before round 4.
// IROTAVG_BCR_UP_CAP: the capacity in 1/1024ths (tests).
template <int B>
static size_t bcr_up_lds(int ntop) {
    const size_t chunk = ((size_t)16 * B * B + 9 * B * 3 + 2) * sizeof(double);
    return ntop > 0 ? std::max(chunk, bcr_top_lds(B, ntop)) : chunk;
}
namespace {
std::atomic<int> g_up_used[16];
template <int B>
int bcr_up_capacity(int dev) {
    static std::atomic<int> cap[16];
    const int d = dev & 15;
    int c = cap[d].load(std::memory_order_relaxed);
    if (c == 0) {
        int per_cu = 0, cus = 0;
        // (dynamic LDS: at least a chunk's; a sixteen-block top asks for more, still one workgroup per CU either way)
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_bcr_reduce_up<B>, 512, bcr_up_lds<B>(0)) != hipSuccess) per_cu = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) cus = 0;
        (void)hipGetLastError();
        c = per_cu > 0 && cus > 0 ? per_cu * cus : -1;
        cap[d].store(c, std::memory_order_relaxed);
    }
    return c;
}
template <int B>
bool bcr_up_reserve(Graph &g, int nwg) {
    const int capwg = bcr_up_capacity<B>(g.device);
    if (capwg <= 0) return false;
    int total = 1024;
    if (const char *e = getenv("IROTAVG_BCR_UP_CAP")) total = atoi(e);
    const int cost = (int)(((long long)nwg * 1024 + capwg - 1) / capwg);
    const int need = cost - g.bcr_up_held;
    if (need <= 0) return true;
    std::atomic<int> &used = g_up_used[g.device & 15];
    const int before = used.fetch_add(need, std::memory_order_acq_rel);
    if (before + need > total) {
        used.fetch_sub(need, std::memory_order_acq_rel);
        return false;
    }
    g.bcr_up_held += need;
    return true;
}
}  // namespace
void bcr_up_release(Graph &g) noexcept {
    if (g.bcr_up_held > 0) {
        g_up_used[g.device & 15].fetch_sub(g.bcr_up_held, std::memory_order_acq_rel);
        g.bcr_up_held = 0;
    }
}

// The three rounds of the way back of one chunk: its W in sW, the solutions of the separator before it and of its own
// separator in sX[0] and sX[8], the other slots zero. Ends with a workgroup barrier. 256 threads. mask: the positions
// wanted (bit i; closed under what a position depends on, bcr_back_mask) -- the others are not computed.
// Three right-hand sides: a unit of work is four rows of one block (sixteen lanes per row of W, three accumulators
// each); the units of a round are dealt to the four waves, so the single block of round 2 and the two of round 1 do not
// leave three / two waves idle.
__device__ __forceinline__ int bcr_back_mask(int m) {
    if (m & 0x05) m |= 0x02;   // 0, 2 need 1
    if (m & 0x50) m |= 0x20;   // 4, 6 need 5
    if (m & 0x36) m |= 0x08;   // 1, 2, 4, 5 need 3
    return m;
}
template <int B, int NR, int NWB = 4>
__device__ __forceinline__ void bcr_back_rounds(const double *sW, double (*sX)[B * NR], bool placed, int kreal, int wave,
                                                int lane, int mask = 0x7f) {
    typedef BcrDim<B, NR> Dm;
    constexpr int WB = B * Dm::NC;
    const int lk = lane >> 4, lp = lane & 15;
#pragma unroll
    for (int rnd = 2; rnd >= 0; rnd--) {
        if (NR == 3) {
            constexpr int UPB = B / 4;                  // units per block
            const int nbk = 4 >> rnd;                   // blocks of the round: 3 | 1 5 | 0 2 4 6
            for (int u = wave; u < nbk * UPB; u += NWB) {
                const int e = u / UPB, it = u - e * UPB;
                int i, a, c;
                if (rnd == 0) {
                    i = 2 * e;
                    a = i - 1;
                    c = i + 1;
                } else if (rnd == 1) {
                    i = 1 + 4 * e;
                    a = e == 0 ? -1 : 3;
                    c = e == 0 ? 3 : 7;
                } else {
                    i = 3;
                    a = -1;
                    c = 7;
                }
                if (!((mask >> i) & 1) || bcr_ridx(placed, kreal, i) < 0) continue;
                const double *Wi = sW + i * WB;
                const double *xa = sX[a + 1], *xcn = sX[c + 1];
                const int k = 4 * it + lk;
                double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int t = 0; t < Dm::NT; t++) {
                    const int cidx = lp + 16 * t;
                    if (cidx < Dm::NC) {
                        const double wv = Wi[k * Dm::NC + cidx];
                        if (cidx < B) {
                            s0 -= wv * xa[cidx * 3 + 0];
                            s1 -= wv * xa[cidx * 3 + 1];
                            s2 -= wv * xa[cidx * 3 + 2];
                        } else if (cidx < 2 * B) {
                            s0 -= wv * xcn[(cidx - B) * 3 + 0];
                            s1 -= wv * xcn[(cidx - B) * 3 + 1];
                            s2 -= wv * xcn[(cidx - B) * 3 + 2];
                        } else {
                            const int q = cidx - 2 * B;
                            s0 += q == 0 ? wv : 0.0;
                            s1 += q == 1 ? wv : 0.0;
                            s2 += q == 2 ? wv : 0.0;
                        }
                    }
                }
                s0 = bcr_row16_sum(s0);
                s1 = bcr_row16_sum(s1);
                s2 = bcr_row16_sum(s2);
                if (lp == 0) {
                    sX[i + 1][k * 3 + 0] = s0;
                    sX[i + 1][k * 3 + 1] = s1;
                    sX[i + 1][k * 3 + 2] = s2;
                }
            }
        } else {
            static_assert(NR == 3 || NWB == 4, "the closures' right-hand sides ride on four-wave workgroups");
            int i = -1, a = -1, c = 7;
            if (rnd == 0) {
                i = 2 * wave;
                a = i - 1;
                c = i + 1;
            } else if (rnd == 1) {
                if (wave < 2) {
                    i = 1 + 4 * wave;
                    a = wave == 0 ? -1 : 3;
                    c = wave == 0 ? 3 : 7;
                }
            } else if (wave == 0) {
                i = 3;
            }
            if (i >= 0 && bcr_ridx(placed, kreal, i) >= 0) {
                const double *Wi = sW + i * WB;
                const double *xa = sX[a + 1], *xcn = sX[c + 1];
                // a lane per (row, right-hand side)
                for (int o = lane; o < B * NR; o += 64) {
                    const int k = o / NR, q = o - NR * k;
                    const double *wr = Wi + k * Dm::NC;
                    double sacc = wr[2 * B + q];
#pragma unroll 4
                    for (int cidx = 0; cidx < B; cidx++)
                        sacc -= wr[cidx] * xa[cidx * NR + q] + wr[B + cidx] * xcn[cidx * NR + q];
                    sX[i + 1][o] = sacc;
                }
            }
        }
        __syncthreads();
    }
}

// K6 inside the way back (run_irls on the plain direct path; Graph::bcr_apply): the threads that write a view's solution
// row also make its step (step_apply) and sum ||x||; the workgroup's sum goes to slot `slot` of the partial array (a
// DOUBLE per workgroup, not a row of four: level 0 at 100k views has 512 workgroups and level 1 another 73, the pinned
// block 512 rows), summed by the host in slot order like k_apply_step's partials.
__device__ __forceinline__ void bcr_apply_sum(double acc, double *__restrict__ part, int slot) {
    __shared__ double sm[16];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < (int)(blockDim.x >> 6); w++) t += sm[w];   // fixed order
        part[slot] = t;
    }
}

// The way back for one chunk: x_7 and the separator before the chunk come from the coarser level. The chunk's
// W (7 blocks of B x (2B + 3)) is staged in LDS first -- every load of the launch is in flight at once; read row
// by row behind the three dependent rounds it cost a memory round trip per four rows.
template <int B, int NR, bool L0>
__global__ __launch_bounds__(256) void k_bcr_back(int nb, int nred, int n, const double *__restrict__ W,
                                                   const double *__restrict__ xc, double *__restrict__ xl,
                                                   double4 *__restrict__ X, double *__restrict__ Z, int zstride,
                                                   int zoff, int nfar, const double *__restrict__ xext, int place,
                                                   double4 *__restrict__ Qap = nullptr, int fap = 0,
                                                   double *__restrict__ part_ap = nullptr, int slot0 = 0) {
    typedef BcrDim<B, NR> Dm;
    constexpr int WB = B * Dm::NC;     // doubles of one W block
    __shared__ double sW[7 * WB];
    __shared__ double sX[9][B * NR];  // slot 0: separator before the chunk; slot j + 1: block j
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int chunk = blockIdx.x;
    const int kreal = nb - chunk * 8 < 8 ? nb - chunk * 8 : 8;
    const bool placed = place && kreal < 8;  // see kBcrPlace
    const int nblk = placed || kreal > 7 ? 7 : kreal;  // W blocks to stage (the separator, position 7, has none)
    // x of the two separators (from the coarser level) and W are requested together: one memory round trip, not two
    constexpr int NQ = (B * NR + 255) / 256;
    double x8[NQ], x0[NQ];
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int e = tid + 256 * q;
        x8[q] = 0.0;
        x0[q] = 0.0;
        if (e < B * NR) {
            x8[q] = xc[(size_t)chunk * B * NR + e];
            if (chunk > 0) x0[q] = xc[(size_t)(chunk - 1) * B * NR + e];
            else if (xext) x0[q] = xext[e];   // a shard: the separator of the rank before it (dist.hip)
        }
    }
    {
        // (every piece of W requested before the first one is put down: a loop of load -> LDS store pairs left it to the
        // compiler how many round trips that became)
        const v2d *__restrict__ src = reinterpret_cast<const v2d *>(W + (size_t)chunk * 7 * WB);
        v2d *dst = reinterpret_cast<v2d *>(sW);
        const int cnt = nblk * WB / 2;  // B is a multiple of 8: WB is even
        constexpr int NVW = (7 * WB / 2 + 255) / 256;
        v2d wreg[NVW];
#pragma unroll
        for (int v = 0; v < NVW; v++) {
            const int e = tid + 256 * v;
            wreg[v] = e < cnt ? __builtin_nontemporal_load(&src[e]) : v2d{0.0, 0.0};
        }
#pragma unroll
        for (int v = 0; v < NVW; v++) {
            const int e = tid + 256 * v;
            if (e < cnt) dst[e] = wreg[v];
        }
    }
    for (int e = tid; e < 7 * B * NR; e += 256) (&sX[1][0])[e] = 0.0;
#pragma unroll
    for (int q = 0; q < NQ; q++) {
        const int e = tid + 256 * q;
        if (e < B * NR) {
            sX[8][e] = x8[q];
            sX[0][e] = x0[q];
        }
    }
    __syncthreads();
    bcr_back_rounds<B, NR>(sW, sX, placed, kreal, wave, lane);
    double acc_ap = 0.0;
    auto put_row = [&](int row, int t) {
        const double *xs = &sX[1][0] + t * NR;
        X[row] = double4{xs[0], xs[1], xs[2], 0.0};
        if (NR > 3)
            for (int q = 0; q < nfar; q++) Z[(size_t)row * zstride + zoff + q] = xs[3 + q];
        if (NR == 3 && Qap) acc_ap += step_apply(xs[0], xs[1], xs[2], Qap, row + fap, true);
    };
    if (L0) {
        const int row0 = chunk * 8 * B;
        for (int t = tid; t < kreal * B; t += 256) {
            const int row = row0 + t, blk = t / B;
            if (row < n) put_row(row, bcr_pos(placed, kreal, blk) * B + (t - blk * B));
        }
    } else {
        for (int e = tid; e < 8 * B * NR; e += 256) {
            const int blk = e / (B * NR);
            xl[(size_t)chunk * 8 * B * NR + e] =
                blk < kreal ? (&sX[1][0])[bcr_pos(placed, kreal, blk) * B * NR + (e - blk * B * NR)] : 0.0;
        }
        // blocks of a mixed level that are level-0 blocks themselves: their solution rows
        if (chunk * 8 + 8 > nred && nred < nb)
            for (int t = tid; t < 8 * B; t += 256) {
                const int i = t / B, r = t - i * B, gb = chunk * 8 + i;
                if (gb >= nred && gb < nb) {
                    const int row = (8 * nred + (gb - nred)) * B + r;
                    if (row < n) put_row(row, t);
                }
            }
    }
    if (NR == 3 && Qap) bcr_apply_sum(acc_ap, part_ap, slot0 + chunk);
}

// The ways back of ALL levels above level 0 in one launch (round 4; they were a launch per level, 11 - 16 us each for
// a handful of workgroups). A workgroup = a chunk of level `base`. What it needs from the level above are the solutions
// of two neighbouring blocks, i.e. the way back of the one or two chunks of that level that hold them, which in turn
// need at most two chunks of the level above them, ... up to the single chunk of the top level: at most two ways back
// per level, computed REDUNDANTLY by every workgroup that needs them -- no workgroup waits for another, nothing is
// exchanged through memory. The solutions of the (<= 16) blocks solved at a level stay in LDS for the level below.
struct BcrBackPlan {
    const double *W[kMaxLevels];
    int nb[kMaxLevels];
    int top, base;
};
// (eight waves: the rounds of a chunk are 24 / 12 / 6 units of four rows, and a workgroup has a CU to itself)
constexpr int kBackTopThreads = 512;
template <int B>
__global__ __launch_bounds__(kBackTopThreads) void k_bcr_back_top(BcrBackPlan P, const double *__restrict__ xtop, double *__restrict__ xl,
                                                       double4 *__restrict__ X, int n, int nred,
                                                       double4 *__restrict__ Qap = nullptr, int fap = 0,
                                                       double *__restrict__ part_ap = nullptr, int slot0 = 0,
                                                       int top_all = 0) {
    // top_all: xtop holds the solutions of ALL blocks of the level above P.top (the single-workgroup top, bcr_top_body,
    // made its own way back): chunk q of level P.top takes blocks q and q - 1 of it
    constexpr int NR = 3;
    typedef BcrDim<B, NR> Dm;
    constexpr int WB = B * Dm::NC, XB = B * NR;
    constexpr int NV = (7 * WB / 2 + kBackTopThreads - 1) / kBackTopThreads;   // 16-byte pieces of a chunk's W per thread
    __shared__ double sW[7 * WB];
    __shared__ double sX[9][XB];
    __shared__ double sWin[2][16][XB];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int cbase = blockIdx.x;
    // the chunks [lo, hi] of level L this workgroup solves (hi - lo <= 1)
    auto range = [&](int L, int &lo, int &hi) {
        lo = hi = cbase;
        for (int l = P.base; l < L; l++) {
            lo = (lo > 0 ? lo - 1 : 0) >> 3;
            hi >>= 3;
        }
    };
    // the positions of chunk q of level L whose solutions are wanted: all at the base level; above it the blocks the
    // chunks of the level below need (the separators of those chunks and the one before the first), plus what they
    // depend on inside the chunk
    auto wanted = [&](int L, int q) {
        if (L == P.base) return 0x7f;
        int lo, hi, m = 0;
        range(L - 1, lo, hi);
        for (int j = (lo > 0 ? lo - 1 : 0); j <= hi; j++)
            if ((j >> 3) == q && (j & 7) < 7) m |= 1 << (j & 7);
        return bcr_back_mask(m);
    };
    // W of the next chunk is requested while this one is solved: pieces of 16 bytes in registers
    v2d wreg[NV];
    auto request = [&](int L, int q) {
        const int kreal = P.nb[L] - q * 8 < 8 ? P.nb[L] - q * 8 : 8;
        const int nblk = kreal > 7 ? 7 : kreal;
        const int mask = wanted(L, q);
        const v2d *__restrict__ src = reinterpret_cast<const v2d *>(P.W[L] + (size_t)q * 7 * WB);
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const int e = tid + kBackTopThreads * v, blk = e / (WB / 2);
            wreg[v] = (blk < nblk && ((mask >> blk) & 1)) ? __builtin_nontemporal_load(&src[e]) : v2d{0.0, 0.0};
        }
    };
    auto deposit = [&]() {
        v2d *dst = reinterpret_cast<v2d *>(sW);
#pragma unroll
        for (int v = 0; v < NV; v++) {
            const int e = tid + kBackTopThreads * v;
            if (e < 7 * WB / 2) dst[e] = wreg[v];
        }
    };
    double acc_ap = 0.0;
    int L = P.top, lo, hi;
    range(L, lo, hi);
    int q = lo;
    request(L, q);
    while (true) {
        int plo = 0, phi = 0;
        if (L < P.top) range(L + 1, plo, phi);
        const double(*winP)[XB] = sWin[(L + 1) & 1];
        double(*winC)[XB] = sWin[L & 1];
        const int kreal = P.nb[L] - q * 8 < 8 ? P.nb[L] - q * 8 : 8;
        const int mask = wanted(L, q);
        deposit();
        for (int e = tid; e < 7 * XB; e += kBackTopThreads) (&sX[1][0])[e] = 0.0;
        for (int e = tid; e < XB; e += kBackTopThreads) {
            sX[8][e] = L == P.top ? xtop[(top_all ? q * XB : 0) + e] : winP[q - 8 * plo][e];
            sX[0][e] = q == 0 ? 0.0 : L == P.top ? (top_all ? xtop[(q - 1) * XB + e] : 0.0) : winP[q - 1 - 8 * plo][e];
        }
        // the next chunk: the same level's second chunk, else the first of the level below
        int Ln = L, qn = q + 1;
        if (qn > hi) {
            Ln = L - 1;
            if (Ln >= P.base) {
                range(Ln, lo, hi);
                qn = lo;
            }
        }
        if (Ln >= P.base) request(Ln, qn);
        __syncthreads();
        bcr_back_rounds<B, NR, kBackTopThreads / 64>(sW, sX, false, kreal, wave, lane, mask);
        if (L > P.base) {
            int clo, chi;
            range(L, clo, chi);
            for (int e = tid; e < 8 * XB; e += kBackTopThreads) {
                const int blk = e / XB;
                winC[(q - clo) * 8 + blk][e - blk * XB] = blk < kreal ? (&sX[1][0])[e] : 0.0;
            }
        } else {
            for (int e = tid; e < 8 * XB; e += kBackTopThreads) {
                const int blk = e / XB;
                xl[(size_t)q * 8 * XB + e] = blk < kreal ? (&sX[1][0])[e] : 0.0;
            }
            // blocks of a mixed level that are level-0 blocks themselves: their solution rows
            if (q * 8 + 8 > nred && nred < P.nb[L])
                for (int t = tid; t < 8 * B; t += kBackTopThreads) {
                    const int i = t / B, r = t - i * B, gb = q * 8 + i;
                    if (gb >= nred && gb < P.nb[L]) {
                        const int row = (8 * nred + (gb - nred)) * B + r;
                        const double *xs = &sX[1][0] + t * NR;
                        if (row < n) {
                            X[row] = double4{xs[0], xs[1], xs[2], 0.0};
                            if (Qap) acc_ap += step_apply(xs[0], xs[1], xs[2], Qap, row + fap, true);
                        }
                    }
                }
        }
        if (Ln < P.base) break;
        __syncthreads();
        L = Ln;
        q = qn;
    }
    if (Qap) bcr_apply_sum(acc_ap, part_ap, slot0 + cbase);
}

// ---------------------------------------------------------------------------------------------
// The closures' part of a solve (round 4). The operator is A = A_b + V C V' (A_b: the band part the reduction
// factorised, column q of V = e_i - e_j for closure q, C = diag(w_q)), so with Y = A_b^-1 b
//     x = Y - A_b^-1 V lambda,    (C^-1 + V' A_b^-1 V) lambda = V' Y        (Sherman-Morrison-Woodbury).
// Round 3 got A_b^-1 V by riding sixteen incidence vectors at a time through a whole re-factorisation (0.3 ms per
// sixteen closures, at most 64). But a column of V has TWO non-zeros, and the block elimination is a nested dissection:
// the forward elimination of e_i only ever touches the blocks on the path from i's block up the elimination tree -- three
// eliminations per level and endpoint, plus what spills into the separator before the chunk. And nothing but the
// forward elimination is needed: with R_g(v) = the right-hand side of block g when it is eliminated,
//     u' A_b^-1 v = sum over the eliminated blocks g of R_g(u)' D_g^-1 R_g(v),
// so S = C^-1 + V' A_b^-1 V and T = V' Y are sums over the blocks two closures' paths share (Y's forward elimination
// is what the reduction stored as W_R), and x = A_b^-1 (b - V lambda) is the ordinary way back from right-hand
// sides corrected by sum_q lambda_q D_g^-1 R_g(v_q) on the blocks of the paths. Which blocks a closure touches
// depends on the positions of its endpoints only: the host writes a STEP PROGRAM per closure once (bcr_closure_plan),
// a wave executes it per solve (k_bcr_closure_forward), pairs of closures are joined on the blocks they share
// (k_bcr_closure_S), the Woodbury system is solved in LDS (<= 96 closures) or by the blocked Gauss-Jordan sweep of
// dense.hip, and one launch corrects the stored right-hand sides (k_bcr_closure_correct) before the ways back.
// No second pass over the matrix, no n x r array: 2048 closures instead of 64.
// ---------------------------------------------------------------------------------------------
struct BcrClPlan {
    const double *W[kMaxLevels], *Dinv[kMaxLevels];
    double *Wrw[kMaxLevels];  // the same W, writable (k_bcr_closure_correct)
    const double *topDinv;
    double *xtop;
};

// One WORKGROUP per closure: its step program. A step eliminates one block g of the path: R = slot[src];
// slot[dstA] -= W_P' R, slot[dstC] -= W_Q' R (the block's W = D^-1 [P' | Q | R_main]), T += R' W_R(main),
// and R and D^-1 R are recorded. The chain of steps is sequential in R (LDS) only: a step's block columns
// ([W_P | W_Q | W_R | D^-1]: 14 KB at B = 24) do not depend on the step before. Round 4 ran a wave per closure that
// fetched its block at every step -- ~40 memory round trips in sequence, 61 us whatever the number of closures; round
// 5: the four waves of the workgroup stage the blocks of FOUR steps in LDS per round trip (each wave one step) and then
// run the four steps from LDS together -- thread (column, part) sums a part of its column's rows, the parts meet in LDS
// (one wave doing a step alone is two chains of B dependent multiply-adds: ~1 us, as long as the round trip it replaced).
template <int B, int NWF = 4>
__global__ __launch_bounds__(64 * NWF) void k_bcr_closure_forward(BcrClPlan P, int r, int nslots, const int *__restrict__ off,
                                                              const int4 *__restrict__ steps,
                                                              const int4 *__restrict__ init, double *__restrict__ recR,
                                                              double *__restrict__ recW, double *__restrict__ T,
                                                              const double *__restrict__ slot_init = nullptr,
                                                              const int2 *__restrict__ fin = nullptr,
                                                              const int *__restrict__ gid = nullptr,
                                                              double *__restrict__ dep = nullptr, int world = 0,
                                                              int rank = 0) {
    constexpr int NR = 3, NC = 2 * B + NR, BB = B * B, NCOL = NC + B, NE = B * NCOL;
    constexpr int NLD = (NE + 63) / 64;
    constexpr int KP = B == 24 ? 3 : (NCOL * 4 <= 256 ? 4 : 2), KB = B / KP;  // parts of a column's rows, rows per part
    static_assert(B % KP == 0 && NCOL * KP <= 256, "thread (col, part)");
    // NWF waves = NWF steps staged per memory round trip (four; eight measured slower, see bcr_launch_forward)
    extern __shared__ double smem[];  // [NWF][B][NCOL] staged blocks, KP x NCOL partial sums, then nslots x B slots
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int q = blockIdx.x;
    double *stage = smem + (size_t)wave * NE;
    double *spart = smem + (size_t)NWF * NE;
    double *slots = spart + KP * NCOL;
    if (wave == 0) {
        if (slot_init) {
            // the separator system of a sharded sequence: the column starts from what the ranks' eliminations left there
            for (int e = lane; e < nslots * B; e += 64) slots[e] = slot_init[(size_t)q * nslots * B + e];
        } else {
            for (int e = lane; e < nslots * B; e += 64) slots[e] = 0.0;
            const int4 in = init[q];
            if (lane == 0) {  // (a slot of 255: that endpoint is a row of another rank)
                if (in.x != 255) slots[in.x * B + in.y] += 1.0;
                if (in.z != 255) slots[in.z * B + in.w] -= 1.0;
            }
        }
    }
    double tacc = 0.0;  // threads 2B .. 2B + 2: a coordinate of T each
    const int s0 = off[q], s1 = off[q + 1];
    __syncthreads();  // (the slots' start values)
    for (int g0 = s0; g0 < s1; g0 += NWF) {
        // ---- every wave: the block of step g0 + wave into its stage, element (k, col) at k NCOL + col ----
        const int st_w = g0 + wave;
        if (st_w < s1) {
            const int4 sp = steps[st_w];
            const int lvl = __builtin_amdgcn_readfirstlane(sp.x), spy = __builtin_amdgcn_readfirstlane(sp.y);
            const double *Wb = lvl >= 0 ? P.W[lvl] + (size_t)spy * B * NC : P.topDinv;
            const double *Db = lvl >= 0 ? P.Dinv[lvl] + (size_t)spy * BB : P.topDinv;
            double v[NLD];
#pragma unroll
            for (int u = 0; u < NLD; u++) {
                const int e = lane + 64 * u;
                const int k = e / NCOL, col = e - k * NCOL;
                // (an address that is valid for every lane; what is not wanted is zeroed by a select below)
                const double *src = Db;
                if (e < NE) {
                    if (col >= NC)
                        src = Db + k * B + (col - NC);
                    else if (lvl >= 0)
                        src = Wb + k * NC + col;
                    else if (col >= 2 * B)
                        src = P.xtop + k * NR + (col - 2 * B);  // the top block: nothing beside it; its part of Y is xtop
                }
                v[u] = *src;
            }
#pragma unroll
            for (int u = 0; u < NLD; u++) {
                const int e = lane + 64 * u;
                const int k = e / NCOL, col = e - k * NCOL;
                const bool zero = lvl < 0 && col < 2 * B;
                if (e < NE) stage[e] = zero ? 0.0 : v[u];
            }
        }
        __syncthreads();
        // ---- all four waves: the (up to) four steps from LDS. Thread (col, part) sums its share of the rows of column
        // col (a wave alone took ~1 us per step: two chains of B dependent multiply-adds), the parts meet in LDS ----
        for (int w = 0; w < NWF && g0 + w < s1; w++) {
            const int st = g0 + w;
            const int spz = __builtin_amdgcn_readfirstlane(steps[st].z);
            const int src = spz & 255, dA = (spz >> 8) & 255, dC = (spz >> 16) & 255;
            const double *blk = smem + (size_t)w * NE;
            const int tcol = threadIdx.x % NCOL, tpart = threadIdx.x / NCOL;
            if (tpart < KP) {
                const double *rs = slots + src * B + tpart * KB;
                double acc = 0.0;
#pragma unroll
                for (int k = 0; k < KB; k++) acc = fma(blk[(tpart * KB + k) * NCOL + tcol], rs[k], acc);
                spart[tpart * NCOL + tcol] = acc;
            }
            if (threadIdx.x < B) recR[(size_t)st * B + threadIdx.x] = slots[src * B + threadIdx.x];
            __syncthreads();
            if (threadIdx.x < NCOL) {
                const int col = threadIdx.x;
                double acc = spart[col];
#pragma unroll
                for (int pp = 1; pp < KP; pp++) acc += spart[pp * NCOL + col];
                if (col < B) {
                    if (dA != 255) slots[dA * B + col] -= acc;
                } else if (col < 2 * B) {
                    if (dC != 255) slots[dC * B + col - B] -= acc;
                } else if (col < NC) {
                    tacc += acc;
                } else {
                    recW[(size_t)st * B + col - NC] = acc;
                }
            }
            __syncthreads();  // (slots and spart before the next step; the stages before the next group)
        }
    }
    // the T threads: column 2B + c is thread 2B + c in every step
    if (threadIdx.x >= 2 * B && threadIdx.x < 2 * B + 3) T[(size_t)q * 3 + (threadIdx.x - 2 * B)] = tacc;
    if (wave != 0) return;
    // a shard: what the column leaves on this rank's separator (block `rank` of the separator system) and on the one
    // before it (block rank - 1) is ADDED to the ranks' common buffer (the loopback's shards run one after the other)
    if (dep) {
        const int2 fn = fin[q];
        double *d = dep + (size_t)gid[q] * world * B;
        if (lane < B) {
            if (fn.x != 255) d[rank * B + lane] += slots[fn.x * B + lane];
            if (fn.y != 255 && rank > 0) d[(rank - 1) * B + lane] += slots[fn.y * B + lane];
        }
    }
}

// The same step programs, one WAVE per closure and the block fetched at every step (round 4's kernel): small in LDS, so
// that a thousand closures run side by side -- the form for many closures, where the machine is filled by their number.
// One wave per closure: its step program. A step eliminates one block g of the path: R = slot[src];
// slot[dstA] -= W_P' R, slot[dstC] -= W_Q' R (the block's W = D^-1 [P' | Q | R_main]), T += R' W_R(main),
// and R and D^-1 R are recorded. Every lane owns one column of [W_P | W_Q | W_R | D^-1] per pass.
template <int B>
__global__ __launch_bounds__(256) void k_bcr_closure_forward_wave(BcrClPlan P, int r, int nslots, const int *__restrict__ off,
                                                              const int4 *__restrict__ steps,
                                                              const int4 *__restrict__ init, double *__restrict__ recR,
                                                              double *__restrict__ recW, double *__restrict__ T,
                                                              const double *__restrict__ slot_init = nullptr,
                                                              const int2 *__restrict__ fin = nullptr,
                                                              const int *__restrict__ gid = nullptr,
                                                              double *__restrict__ dep = nullptr, int world = 0,
                                                              int rank = 0) {
    constexpr int NR = 3, NC = 2 * B + NR, BB = B * B, NCOL = NC + B;
    extern __shared__ double smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = blockIdx.x * (blockDim.x >> 6) + wave;
    if (q >= r) return;  // (no workgroup barrier below)
    double *slots = smem + (size_t)wave * nslots * B;
    if (slot_init) {
        // the separator system of a sharded sequence: the column starts from what the ranks' eliminations left there
        for (int e = lane; e < nslots * B; e += 64) slots[e] = slot_init[(size_t)q * nslots * B + e];
    } else {
        for (int e = lane; e < nslots * B; e += 64) slots[e] = 0.0;
        const int4 in = init[q];
        if (lane == 0) {  // (a slot of 255: that endpoint is a row of another rank)
            if (in.x != 255) slots[in.x * B + in.y] += 1.0;
            if (in.z != 255) slots[in.z * B + in.w] -= 1.0;
        }
    }
    double tacc = 0.0;  // lanes 2B .. 2B + 2 of the first pass: a coordinate of T each
    for (int st = off[q]; st < off[q + 1]; st++) {
        const int4 sp = steps[st];
        const int lvl = sp.x, src = sp.z & 255, dA = (sp.z >> 8) & 255, dC = (sp.z >> 16) & 255;
        const double *Wb = lvl >= 0 ? P.W[lvl] + (size_t)sp.y * B * NC : nullptr;
        const double *Db = lvl >= 0 ? P.Dinv[lvl] + (size_t)sp.y * BB : P.topDinv;
        double R[B];
        {
            const v2d *rs = reinterpret_cast<const v2d *>(slots + src * B);
#pragma unroll
            for (int k = 0; k < B / 2; k++) {
                const v2d v = rs[k];
                R[2 * k] = v.x;
                R[2 * k + 1] = v.y;
            }
        }
        if (lane < B) recR[(size_t)st * B + lane] = slots[src * B + lane];
        // the columns of both passes are requested before the first is used: one memory round trip per step
        constexpr int NPASS = (NCOL + 63) / 64;
        double m[NPASS][B];
        bool have[NPASS];
#pragma unroll
        for (int pass = 0; pass < NPASS; pass++) {
            const int col = pass * 64 + lane;
            const double *base = Db;
            int stride = B;
            have[pass] = col < NCOL;
            if (col < NC) {
                if (lvl >= 0) {
                    base = Wb + col;
                    stride = NC;
                } else {  // the top block: nothing beside it; its part of Y is xtop
                    base = P.xtop + (col - 2 * B);
                    stride = NR;
                    have[pass] = col >= 2 * B;
                }
            } else {
                base = Db + (col - NC);
            }
#pragma unroll
            for (int k = 0; k < B; k++) m[pass][k] = have[pass] ? base[(size_t)k * stride] : 0.0;
        }
#pragma unroll
        for (int pass = 0; pass < NPASS; pass++) {
            const int col = pass * 64 + lane;
            if (col >= NCOL) continue;
            double acc = 0.0;
#pragma unroll
            for (int k = 0; k < B; k++) acc = fma(m[pass][k], R[k], acc);
            if (col < B) {
                if (dA != 255) slots[dA * B + col] -= acc;
            } else if (col < 2 * B) {
                if (dC != 255) slots[dC * B + col - B] -= acc;
            } else if (col < NC) {
                tacc += acc;
            } else {
                recW[(size_t)st * B + col - NC] = acc;
            }
        }
    }
    // the T lanes: column 2B + c sits in pass (2B + c) / 64, lane (2B + c) % 64 -- the same lane in every step
    {
        const int c0 = 2 * B;
        for (int c = 0; c < 3; c++)
            if (((c0 + c) & 63) == lane) T[(size_t)q * 3 + c] = tacc;
    }
    // a shard: what the column leaves on this rank's separator (block `rank` of the separator system) and on the one
    // before it (block rank - 1) is ADDED to the ranks' common buffer (the loopback's shards run one after the other)
    if (dep) {
        const int2 fn = fin[q];
        double *d = dep + (size_t)gid[q] * world * B;
        if (lane < B) {
            if (fn.x != 255) d[rank * B + lane] += slots[fn.x * B + lane];
            if (fn.y != 255 && rank > 0) d[(rank - 1) * B + lane] += slots[fn.y * B + lane];
        }
    }
}

// S = C^-1 + V' A_b^-1 V: entry (p, q) = sum over the steps of p and q on the same block of R_p . (D^-1 R_q); the
// step lists are sorted by block. A workgroup per 16 x 16 tile of the upper triangle, the keys of its 32 closures in
// LDS (a merge that fetched every key from memory was a dependent load per step: 270 us at a thousand closures); a
// closure of weight 0 (Talwar) is not there: its row and column are the identity's, its T is 0.
template <int B>
__global__ __launch_bounds__(256) void k_bcr_closure_S(int r, int npad, int maxsteps, int ndense,
                                                        const int *__restrict__ dense, const int *__restrict__ off,
                                                        const int4 *__restrict__ steps, const double *__restrict__ recR,
                                                        const double *__restrict__ recW, const int *__restrict__ far_e,
                                                        const double *__restrict__ wsrc, int wsquare,
                                                        double *__restrict__ S, double *__restrict__ T,
                                                        int *__restrict__ alive, const int *__restrict__ gid = nullptr,
                                                        const uint8_t *__restrict__ own = nullptr, int npadG = 0,
                                                        double *__restrict__ Tx = nullptr,
                                                        double *__restrict__ deadx = nullptr) {
    extern __shared__ int skey[];  // [32][maxsteps]: rows 0..15 the tile's p, 16..31 its q
    constexpr int DB = 4;          // dense eliminations per batch
    __shared__ double sx[DB][16][B + 1], sy[DB][16][B + 1];
    if (blockIdx.x < blockIdx.y) return;
    const int tp = threadIdx.x >> 4, tq = threadIdx.x & 15;
    const int p = blockIdx.y * 16 + tp, q = blockIdx.x * 16 + tq;
    for (int e = threadIdx.x; e < 32 * maxsteps; e += 256) {
        const int who = e / maxsteps, k = e - who * maxsteps;
        const int cl = who < 16 ? blockIdx.y * 16 + who : blockIdx.x * 16 + who - 16;
        int key = 0x7fffffff;
        if (cl < r && k < off[cl + 1] - off[cl]) key = steps[off[cl] + k].w;
        skey[e] = key;
    }
    __syncthreads();
    // the dense eliminations (the blocks of the top levels, on almost every closure's path): the records of the tile's
    // sixteen p and sixteen q through LDS -- each is used sixteen times. DB eliminations per batch: their records are
    // requested together (one elimination per round was one memory round trip per elimination: 36 of this kernel's
    // 51 us at a hundred closures).
    double sum = 0.0;
    {
        constexpr int NE = 32 * (B / 2), NU = (DB * NE + 255) / 256;  // pairs of doubles per elimination; per thread and batch
        for (int d0 = 0; d0 < ndense; d0 += DB) {
            v2d v[NU];
#pragma unroll
            for (int u = 0; u < NU; u++) {
                const int e = threadIdx.x + 256 * u;
                const int dd = e / NE, e1 = e - dd * NE;
                const int who = e1 / (B / 2), k2 = e1 - who * (B / 2);
                const int cl = who < 16 ? blockIdx.y * 16 + who : blockIdx.x * 16 + who - 16;
                int st = -1;
                if (dd < DB && d0 + dd < ndense && cl < r) st = dense[(size_t)(d0 + dd) * r + cl];
                v[u] = v2d{0.0, 0.0};
                if (st >= 0) v[u] = *reinterpret_cast<const v2d *>((who < 16 ? recR : recW) + (size_t)st * B + 2 * k2);
            }
            __syncthreads();  // (the previous batch has been read)
#pragma unroll
            for (int u = 0; u < NU; u++) {
                const int e = threadIdx.x + 256 * u;
                const int dd = e / NE, e1 = e - dd * NE;
                const int who = e1 / (B / 2), k2 = e1 - who * (B / 2);
                if (dd < DB) {
                    double *dst = who < 16 ? &sx[dd][who][2 * k2] : &sy[dd][who - 16][2 * k2];
                    dst[0] = v[u].x;
                    dst[1] = v[u].y;
                }
            }
            __syncthreads();
#pragma unroll
            for (int dd = 0; dd < DB; dd++) {
                if (d0 + dd >= ndense) break;
                const double *x = sx[dd][tp], *y = sy[dd][tq];
                double d0s = 0.0, d1s = 0.0;
#pragma unroll
                for (int k = 0; k < B; k += 2) {
                    d0s = fma(x[k], y[k], d0s);
                    d1s = fma(x[k + 1], y[k + 1], d1s);
                }
                sum += d0s + d1s;
            }
        }
    }
    if (p >= npad || q >= npad || q < p) return;
    if (p >= r || q >= r) {  // padding of the inversion
        if (!gid) S[(size_t)p * npad + q] = S[(size_t)q * npad + p] = p == q ? 1.0 : 0.0;
        return;
    }
    double wp = wsrc[far_e[p]], wq = wsrc[far_e[q]];
    if (wsquare) {
        wp *= wp;
        wq *= wq;
    }
    if (wp > 0.0 && wq > 0.0) {
        const int *ka = skey + tp * maxsteps, *kb = skey + (16 + tq) * maxsteps;
        const int na = off[p + 1] - off[p], nb = off[q + 1] - off[q];
        const double *x0 = recR + (size_t)off[p] * B, *y0 = recW + (size_t)off[q] * B;
        int a = 0, b = 0;
        while (a < na && b < nb) {
            const int va = ka[a], vb = kb[b];
            if (va == vb) {
                if (!(va & 1)) {  // (a dense elimination: summed above)
                    const double *x = x0 + (size_t)a * B, *y = y0 + (size_t)b * B;
                    double d0 = 0.0, d1 = 0.0;
#pragma unroll
                    for (int k = 0; k < B; k += 2) {
                        d0 = fma(x[k], y[k], d0);
                        d1 = fma(x[k + 1], y[k + 1], d1);
                    }
                    sum += d0 + d1;
                }
                a++;
                b++;
            } else if (va < vb) {
                a++;
            } else {
                b++;
            }
        }
        if (p == q && (!gid || own[p])) sum += 1.0 / wp;
    } else {
        sum = p == q ? 1.0 : 0.0;
    }
    if (gid) {
        // a shard: this rank's share of the system is ADDED at the closures' places in the global list (dist.hip sums
        // the ranks' buffers); a closure of weight 0 is flagged by its owner and settled after the sum
        const int gp = gid[p], gq = gid[q];
        const bool live = wp > 0.0 && wq > 0.0;
        if (live) {
            S[(size_t)gp * npadG + gq] += sum;
            if (p != q) S[(size_t)gq * npadG + gp] += sum;
        }
        if (p == q) {
            if (live) {
                for (int c = 0; c < 3; c++) Tx[3 * gp + c] += T[3 * p + c];
            } else if (own[p]) {
                deadx[gp] += 1.0;
            }
        }
        return;
    }
    S[(size_t)p * npad + q] = sum;
    S[(size_t)q * npad + p] = sum;
    if (p == q) {
        alive[p] = wp > 0.0;
        if (!(wp > 0.0)) T[3 * p] = T[3 * p + 1] = T[3 * p + 2] = 0.0;
    }
}

// The same sums for FEW closures (round 6), a WAVE per pair (p, q), p <= q. With a tile of 16 x 16 pairs per workgroup
// thirty closures are three workgroups -- three CUs of 256 -- and every thread walks its pair's merge alone, 24 doubles of
// two records per shared block out of global memory: 34 us of an IRLS iteration that has thirty closures, most of it
// latency. Here the wave looks every key of p's list up in q's list (a binary search in LDS, 64 keys at a time), lists
// the matches, and eight groups of eight lanes take a match each per round -- lane e of a group the elements e, e + 8, ...
// of the two records -- so that eight matches' loads are in flight together; one sum over the wave at the end. All
// eliminations alike: the dense ones (the top levels' blocks, an odd key) need no table here. Same entries as
// k_bcr_closure_S up to the order of the additions; single GPU only (a shard adds into the ranks' common buffer).
constexpr int kClosurePairsMax = 192;  // closures up to which the pair kernel is used (above: the tiles' shared keys pay)
template <int B>
__global__ __launch_bounds__(256) void k_bcr_closure_S_pairs(int r, int npad, int maxsteps, const int *__restrict__ off,
                                                              const int4 *__restrict__ steps, const double *__restrict__ recR,
                                                              const double *__restrict__ recW, const int *__restrict__ far_e,
                                                              const double *__restrict__ wsrc, int wsquare,
                                                              double *__restrict__ S, double *__restrict__ T,
                                                              int *__restrict__ alive) {
    extern __shared__ int spairs[];  // per wave: maxsteps keys of q, then 64 matches (a | b << 16)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int p = blockIdx.y, q = blockIdx.x * 4 + wave;
    if (q < p || q >= npad || p >= npad) return;   // (whole waves leave: no barrier below)
    if (p >= r || q >= r) {  // padding of the inversion
        if (lane == 0) S[(size_t)p * npad + q] = S[(size_t)q * npad + p] = p == q ? 1.0 : 0.0;
        return;
    }
    double wp = wsrc[far_e[p]], wq = wsrc[far_e[q]];
    if (wsquare) {
        wp *= wp;
        wq *= wq;
    }
    double sum = 0.0;
    if (wp > 0.0 && wq > 0.0) {
        int *kb = spairs + (size_t)wave * (maxsteps + 64);
        int *ml = kb + maxsteps;
        const int na = off[p + 1] - off[p], nb = off[q + 1] - off[q];
        for (int k = lane; k < nb; k += 64) kb[k] = steps[off[q] + k].w;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        const double *x0 = recR + (size_t)off[p] * B, *y0 = recW + (size_t)off[q] * B;
        const int grp = lane >> 3, e = lane & 7;
        double acc = 0.0;
        for (int a0 = 0; a0 < na; a0 += 64) {
            const int a = a0 + lane;
            int found = -1;
            if (a < na) {
                const int ka = steps[off[p] + a].w;
                int lo = 0, hi = nb;  // first index whose key is >= ka
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (kb[mid] < ka) lo = mid + 1;
                    else hi = mid;
                }
                if (lo < nb && kb[lo] == ka) found = lo;
            }
            const unsigned long long mask = __ballot(found >= 0);
            const int nm = __popcll(mask);
            if (found >= 0) ml[__popcll(mask & ((1ull << lane) - 1ull))] = a | (found << 16);
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            for (int i = grp; i < nm; i += 8) {
                const int ab = ml[i];
                const double *x = x0 + (size_t)(ab & 0xffff) * B, *y = y0 + (size_t)(ab >> 16) * B;
#pragma unroll
                for (int k = 0; k < (B + 7) / 8; k++)
                    if (e + 8 * k < B) acc = fma(x[e + 8 * k], y[e + 8 * k], acc);
            }
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
        sum = acc;
        if (p == q) sum += 1.0 / wp;
    } else {
        sum = p == q ? 1.0 : 0.0;
    }
    if (lane != 0) return;
    S[(size_t)p * npad + q] = sum;
    S[(size_t)q * npad + p] = sum;
    if (p == q) {
        alive[p] = wp > 0.0;
        if (!(wp > 0.0)) T[3 * p] = T[3 * p + 1] = T[3 * p + 2] = 0.0;
    }
}

// lambda = S^-1 T for at most 96 closures: one workgroup, [S | T] in LDS, Gauss-Jordan without pivoting (SPD). Thread
// (row p, class c) updates the columns q = c, c + 4, ... BEHIND the pivot of its row: the row's factor is formed once,
// nobody writes the pivot's row or column during its step, so ONE barrier per pivot. (Round 4's form -- an element per
// thread found by an integer division, three barriers per pivot, every column -- took 225 us at sixty closures, a third
// of such an IRLS iteration and more than the blocked sweep of dense.hip needs for a hundred: round 5, ~25 us; the time
// grows with r^2 and meets the blocked sweep's 92 us at about a hundred closures.)
constexpr int kSolveLdsMax = 96;  // (measured at 100k views: the blocked sweep of dense.hip takes over at ~ 100 closures: 92 us there)
__global__ __launch_bounds__(512) void k_bcr_closure_solve_lds(int r, int npad, const double *__restrict__ Sg,
                                                                const double *__restrict__ Tg, double *__restrict__ lam) {
    extern __shared__ double A[];  // r x ld
    const int nc = r + 3, ld = nc | 1;
    const int tid = threadIdx.x, p = tid >> 2, c = tid & 3;
    for (int e = tid; e < r * nc; e += blockDim.x) {
        const int pp = e / nc, q = e - pp * nc;
        A[pp * ld + q] = q < r ? Sg[(size_t)pp * npad + q] : Tg[3 * pp + (q - r)];
    }
    __syncthreads();
    for (int k = 0; k < r; k++) {
        if (p < r && p != k) {
            const double f = A[p * ld + k] / A[k * ld + k];
            const double *rk = A + k * ld;
            double *rp = A + p * ld;
            for (int q = k + 1 + ((c - (k + 1)) & 3); q < nc; q += 4) rp[q] = fma(-f, rk[q], rp[q]);
        }
        __syncthreads();
    }
    if (p < r && c < 3) lam[3 * p + c] = A[p * ld + r + c] / A[p * ld + p];
}
static void bcr_launch_solve_lds(hipStream_t st, int r, int npad, const double *S, const double *T, double *lam) {
    const size_t lds = (size_t)r * ((r + 3) | 1) * sizeof(double);
    static std::atomic<size_t> lds_set[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (lds > 64 * 1024 && lds_set[dev & 15].load() < lds) {
        IRH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bcr_closure_solve_lds),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        lds_set[dev & 15].store(lds);
    }
    hipLaunchKernelGGL(k_bcr_closure_solve_lds, dim3(1), dim3(r <= 64 ? 256 : 512), lds, st, r, npad, S, T, lam);
}

// lambda = Sinv T (more than 96 closures: Sinv from the blocked Gauss-Jordan sweep); a wave per row
__global__ __launch_bounds__(256) void k_bcr_closure_lambda(int r, int npad, const double *__restrict__ Sinv,
                                                             const double *__restrict__ T, double *__restrict__ lam) {
    const int lane = threadIdx.x & 63, p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= r) return;
    double s0 = 0.0, s1 = 0.0, s2 = 0.0;
    for (int q = lane; q < r; q += 64) {
        const double v = Sinv[(size_t)p * npad + q];
        s0 = fma(v, T[3 * q], s0);
        s1 = fma(v, T[3 * q + 1], s1);
        s2 = fma(v, T[3 * q + 2], s2);
    }
    s0 = wave_sum(s0);
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) {
        lam[3 * p] = s0;
        lam[3 * p + 1] = s1;
        lam[3 * p + 2] = s2;
    }
}

// W_R of every block a closure touches -= sum over the steps on it of (D^-1 R_q) lambda_q' (the top block: xtop):
// a workgroup per block. The top blocks lie on every closure's path (a thousand steps each): the list is dealt to
// PARTS groups of threads whose partial sums are added in a fixed order.
template <int B>
__global__ __launch_bounds__(1024) void k_bcr_closure_correct(BcrClPlan P, const int *__restrict__ eoff,
                                                               const int *__restrict__ erec, const int2 *__restrict__ elim,
                                                               const int *__restrict__ owner,
                                                               const double *__restrict__ recW,
                                                               const double *__restrict__ lam) {
    constexpr int NR = 3, NC = 2 * B + NR, NO = B * 3, PARTS = 1024 / NO;
    __shared__ double part[PARTS][NO];
    const int g = blockIdx.x, t = threadIdx.x;
    const int pt = t / NO, o = t - pt * NO;
    const int row = o / 3, c = o - 3 * row;
    const int k0 = eoff[g], k1 = eoff[g + 1];
    if (pt < PARTS) {
        double acc = 0.0;
        if (k1 - k0 > pt) {
#pragma unroll 4
            for (int k = k0 + pt; k < k1; k += PARTS) {
                const int st = erec[k];
                acc = fma(recW[(size_t)st * B + row], lam[3 * owner[st] + c], acc);
            }
        }
        part[pt][o] = acc;
    }
    __syncthreads();
    if (t >= NO) return;
    double acc = 0.0;
    const int np = k1 - k0 < PARTS ? k1 - k0 : PARTS;
    for (int q = 0; q < np; q++) acc += part[q][o];
    const int2 el = elim[g];
    if (el.x >= 0)
        P.Wrw[el.x][(size_t)el.y * B * NC + (size_t)row * NC + 2 * B + c] -= acc;
    else
        P.xtop[row * NR + c] -= acc;
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// The step programs of the closures (see "The closures' part of a solve"): a symbolic forward elimination of
// e_i - e_j per closure. Blocks are tracked as (level, block of that level) -> slot; a chunk's eliminations run in the
// kernel's order (0 2 4 6)(1 5)(3); what an elimination subtracts from the separator before its chunk is kept in the
// slot of that separator's block of the NEXT level (the reduction adds sepR[j] + extR[j + 1] there as well).
// Where the real block t of a chunk with kreal blocks sits (host mirror of bcr_pos / kBcrPlace)
static const signed char kBcrPlaceHost[9][8] = {{-1, -1, -1, -1, -1, -1, -1, -1}, {7, -1, -1, -1, -1, -1, -1, -1},
                                                {3, 7, -1, -1, -1, -1, -1, -1},   {3, 5, 7, -1, -1, -1, -1, -1},
                                                {1, 3, 5, 7, -1, -1, -1, -1},     {1, 3, 5, 6, 7, -1, -1, -1},
                                                {1, 3, 4, 5, 6, 7, -1, -1},       {1, 2, 3, 4, 5, 6, 7, -1},
                                                {0, 1, 2, 3, 4, 5, 6, 7}};
// A SHARD's plan (round 5): the partial chunks are placed (the last block sits at position 7 and survives as the
// separator at every level), chunk 0 of every level subtracts from the separator of the rank before this one (one slot
// per closure that lives through all levels), the last level is not solved -- what is left at its position 7 and in that
// slot goes to the separator system (cl_fin), and an endpoint may be a row of another rank (row -1: no start value here).
static void bcr_closure_plan(Graph &g) {
    BcrState &S = *g.bcr;
    const int B = S.B, nl = (int)S.lev.size(), r = S.nfar;
    const int nch0 = S.lev[0].nch;
    const bool shard = g.bcr_shard;
    std::vector<int> off(1, 0), owner;
    std::vector<int4> steps, init((size_t)r);
    std::vector<int2> fin((size_t)r);
    std::vector<long long> koff((size_t)nl + 1, 0);  // sort keys: (level, chunk, rank in the schedule)
    for (int l = 0; l < nl; l++) koff[(size_t)l + 1] = koff[(size_t)l] + (long long)S.lev[l].nch * 8;
    static const int rank_of[7] = {0, 4, 1, 6, 2, 5, 3}, sched[7] = {0, 2, 4, 6, 1, 5, 3};
    int nslots = 2;
    for (int q = 0; q < r; q++) {
        std::vector<std::map<int, int>> nz((size_t)nl + 1);  // per level: block -> slot
        std::vector<int> freelist;
        int next = 0;
        auto take = [&]() {
            if (!freelist.empty()) {
                const int t = freelist.back();
                freelist.pop_back();
                return t;
            }
            return next++;
        };
        auto slot_of = [&](int l, int blk) {
            auto it = nz[(size_t)l].find(blk);
            if (it != nz[(size_t)l].end()) return it->second;
            const int t = take();
            nz[(size_t)l][blk] = t;
            return t;
        };
        int4 in;
        in.x = in.z = 255;
        in.y = in.w = 0;
        for (int e = 0; e < 2; e++) {
            const int row = e ? g.bcr_far_j[(size_t)q] : g.bcr_far_i[(size_t)q];
            if (row < 0) continue;  // (a row of another rank)
            const int b = row / B;
            int l = 0, blk = b;
            if (nl > 1 && b >= 8 * nch0) {  // a block no chunk of level 0 reduces: a block of the mixed level 1
                l = 1;
                blk = S.lev[1].nred + (b - 8 * nch0);
            }
            const int t = slot_of(l, blk);
            if (e == 0) {
                in.x = t;
                in.y = row - b * B;
            } else {
                in.z = t;
                in.w = row - b * B;
            }
        }
        init[(size_t)q] = in;
        int remote = -1, own = -1;  // a shard: the slots of the two separators this rank's eliminations add to
        for (int l = 0; l < nl; l++) {
            const bool top = l == nl - 1 && !shard;
            // chunks with non-zero blocks, ascending (std::map is ordered)
            while (!nz[(size_t)l].empty()) {
                const int chunk = nz[(size_t)l].begin()->first / 8;
                const int kreal = std::min(8, S.lev[l].nb - chunk * 8);
                const bool placed = shard && kreal < 8;
                auto pos_of = [&](int t) { return placed ? (int)kBcrPlaceHost[kreal][t] : t; };
                bool real[8];
                for (int i = 0; i < 8; i++) real[i] = false;
                for (int t = 0; t < kreal; t++) real[pos_of(t)] = true;
                int slot[9];  // positions 0..7, [8] = the separator before the chunk
                for (int i = 0; i < 9; i++) slot[i] = -1;
                for (int t = 0; t < kreal; t++) {
                    auto it = nz[(size_t)l].find(chunk * 8 + t);
                    if (it != nz[(size_t)l].end()) {
                        slot[pos_of(t)] = it->second;
                        nz[(size_t)l].erase(it);
                    }
                }
                const bool ext_remote = shard && chunk == 0 && S.ext0;
                const bool has_ext = chunk > 0 || ext_remote;
                for (int k = 0; k < 7; k++) {
                    const int i = sched[k];
                    if (!real[i] || slot[i] < 0) continue;
                    int a, c;
                    if (k < 4) {
                        a = i - 1;
                        c = i + 1;
                    } else if (k < 6) {
                        a = i == 1 ? -1 : 3;
                        c = i == 1 ? 3 : 7;
                    } else {
                        a = -1;
                        c = 7;
                    }
                    int dA = 255, dC = 255;
                    if (a >= 0) {
                        // (a placed chunk: every padding position has been skipped before it would be a neighbour)
                        if (!real[a]) throw HipError{hipErrorUnknown};
                        if (slot[a] < 0) slot[a] = take();
                        dA = slot[a];
                    } else if (has_ext) {
                        if (slot[8] < 0) {
                            if (ext_remote) {
                                if (remote < 0) remote = take();
                                slot[8] = remote;
                            } else {
                                // the separator before the chunk = block chunk - 1 of the next level (it may hold a value)
                                slot[8] = slot_of(l + 1, chunk - 1);
                            }
                        }
                        dA = slot[8];
                    }
                    if (real[c]) {
                        if (slot[c] < 0) slot[c] = take();
                        dC = slot[c];
                    }
                    int4 sp;
                    sp.x = l;
                    sp.y = chunk * 7 + i;
                    sp.z = slot[i] | (dA << 8) | (dC << 16);
                    sp.w = (int)(koff[(size_t)l] + (long long)chunk * 8 + rank_of[i]);
                    steps.push_back(sp);
                    owner.push_back(q);
                    slot[i] = -1;  // (its slot is not reused: a new slot must start from zero)
                }
                // what is left: the separator (position 7 when the chunk is full or placed)
                if (real[7] && slot[7] >= 0) {
                    if (top) {  // solved on the spot
                        int4 sp;
                        sp.x = -1;
                        sp.y = 0;
                        sp.z = slot[7] | (255 << 8) | (255 << 16);
                        sp.w = (int)(koff[(size_t)l] + (long long)chunk * 8 + 7);
                        steps.push_back(sp);
                        owner.push_back(q);
                    } else if (l == nl - 1) {  // a shard's last level: this rank's block of the separator system
                        own = slot[7];
                    } else {
                        auto it = nz[(size_t)l + 1].find(chunk);
                        if (it == nz[(size_t)l + 1].end()) {
                            nz[(size_t)l + 1][chunk] = slot[7];
                        } else {
                            // the block of the next level already has a slot (the chunk after this one subtracted from
                            // it first): cannot happen -- chunks run in ascending order and only chunk + 1 writes there
                            throw HipError{hipErrorUnknown};
                        }
                    }
                }
            }
        }
        fin[(size_t)q] = int2{own < 0 ? 255 : own, remote < 0 ? 255 : remote};
        nslots = std::max(nslots, next);
        S.cl_maxsteps = std::max(S.cl_maxsteps, (int)steps.size() - off.back());
        off.push_back((int)steps.size());
    }
    if (nslots > 250) throw HipError{hipErrorUnknown};
    // the steps of every touched elimination, in step order
    std::map<int, std::vector<int>> by_key;
    for (size_t k = 0; k < steps.size(); k++) by_key[steps[k].w].push_back((int)k);
    std::vector<int> eoff(1, 0), erec;
    std::vector<int2> elim;
    for (auto &kv : by_key) {
        const int4 sp = steps[(size_t)kv.second[0]];
        int2 el;
        el.x = sp.x;
        el.y = sp.y;
        elim.push_back(el);
        for (int k : kv.second) erec.push_back(k);
        eoff.push_back((int)erec.size());
    }
    // eliminations on (almost) every closure's path -- the blocks of the top levels: their part of S is a dense
    // product, done tile-wise from LDS (k_bcr_closure_S); the key of a step carries the mark in its lowest bit
    std::vector<int> dense;
    for (size_t k = 0; k < steps.size(); k++) steps[k].w *= 2;
    if (r >= 8) {  // (round 5: also for the Woodbury systems that are solved in LDS -- the merge is a memory round trip per match)
        int d = 0;
        for (auto &kv : by_key)
            if ((long long)kv.second.size() * 4 >= r) {
                dense.resize((size_t)(d + 1) * r, -1);
                for (int k : kv.second) {
                    dense[(size_t)d * r + owner[(size_t)k]] = k;
                    steps[(size_t)k].w |= 1;
                }
                d++;
            }
        S.cl_ndense = d;
    }
    if (dense.empty()) dense.push_back(-1);
    S.cl_dense.upload(dense, g.stream);
    S.cl_nslots = nslots;
    S.cl_nsteps = (int)steps.size();
    S.cl_nelim = (int)elim.size();
    S.cl_npad = r <= kSolveLdsMax ? r : (r + 63) / 64 * 64;
    S.far_i.upload(g.bcr_far_i, g.stream);
    S.far_j.upload(g.bcr_far_j, g.stream);
    S.far_e.upload(g.bcr_far_e, g.stream);
    S.cl_off.upload(off, g.stream);
    S.cl_step.upload(steps, g.stream);
    S.cl_init.upload(init, g.stream);
    if (shard) {
        // (the steps' owners as the correction reads them: numbers in the global list, where lambda lives)
        std::vector<int> gowner(owner.size());
        for (size_t k = 0; k < owner.size(); k++) gowner[k] = g.bcr_far_gid[(size_t)owner[k]];
        if (gowner.empty()) gowner.push_back(0);
        S.cl_owner.upload(gowner, g.stream);
        S.cl_fin.upload(fin, g.stream);
        S.cl_gid.upload(g.bcr_far_gid, g.stream);
        S.cl_own.upload(g.bcr_far_own, g.stream);
    } else {
        S.cl_owner.upload(owner, g.stream);
    }
    S.co_off.upload(eoff, g.stream);
    S.co_rec.upload(erec, g.stream);
    S.co_elim.upload(elim, g.stream);
    S.cl_R.alloc(steps.size() * (size_t)B);
    S.cl_W.alloc(steps.size() * (size_t)B);
    if (!shard) S.cl_S.alloc((size_t)S.cl_npad * S.cl_npad);  // (a shard's share goes straight into the ranks' common buffer)
    S.cl_T.alloc((size_t)S.cl_npad * 3);
    S.lam.alloc((size_t)S.cl_npad * 3);
    S.cl_alive.alloc((size_t)S.cl_npad);
    S.Dinv.resize((size_t)nl);
    for (int l = 0; l < nl; l++) S.Dinv[(size_t)l].alloc((size_t)S.lev[l].nch * 7 * B * B);
    S.topDinv.alloc((size_t)B * B);
    S.dead.alloc(1);
    IRH_CHECK(hipMemsetAsync(S.topDinv.p, 0, sizeof(double) * (size_t)B * B, g.stream));
    IRH_CHECK(hipStreamSynchronize(g.stream));  // the host vectors go away
}

static void bcr_alloc(Graph &g) {
    if (g.bcr) return;
    g.bcr.reset(new BcrState());
    BcrState &S = *g.bcr;
    S.B = g.bcr_B;
    S.nfar = (int)g.bcr_far_e.size();
    S.NR = 3;
    const int B = S.B, NR = S.NR, NC = 2 * B + NR;
    const int nb0 = (g.levels[0].n + B - 1) / B;
    int nch0 = (nb0 + 7) / 8;
    // The level-0 reduction keeps two workgroups per CU resident (one for B = 32 or with closure columns: LDS). A
    // chunk count slightly above a multiple of that capacity would cost a whole extra round of workgroups for a
    // handful of chunks (100k views, B = 24: 521 chunks on 512 slots): the chunks beyond the multiple are not
    // reduced at all -- their blocks enter level 1 as they are (a MIXED level 1: k_bcr_reduce's loader), where
    // there is room.
    int nraw = 0;
    {
        int ncu = 256;
        (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, g.device);
        if (ncu <= 0) ncu = 256;
        const int cap = ncu * (B <= 24 && NR == 3 ? 2 : 1);
        const int full = nch0 / cap * cap, rem = nch0 - full;
        if (full > 0 && rem > 0 && rem <= cap / 4 && !g.bcr_shard && !getenv("IROTAVG_BCR_NO_MIXED")) {
            nraw = nb0 - 8 * full;
            nch0 = full;
        }
    }
    // the last level: one chunk of <= 8 blocks, or (round 5) one workgroup of <= 16 blocks that also makes its way back
    // -- three right-hand sides, blocks up to 24 rows (LDS), no closures (their step programs know chunks), not a shard
    S.top16 = B <= 24 && S.nfar == 0 && !g.bcr_shard && !getenv("IROTAVG_BCR_NO_TOP16");
    const int topmax = S.top16 ? kTopMax : 8;
    int nb = nb0, nch = nch0;
    for (int l = 0;; l++) {
        S.lev.emplace_back();
        BcrLevel &L = S.lev.back();
        L.nb = nb;
        L.nch = nch;
        L.nred = (l == 1 && nraw > 0) ? nb - nraw : nb;
        const bool last = nb <= topmax && !g.bcr_shard && l > 0;
        if (last && S.top16) {
            // (a level 0 of <= 16 blocks keeps the chunk form: its blocks are gathered from the operator)
            L.x.alloc((size_t)L.nch * 8 * B * NR);
            S.tsched = bcr_top_schedule(nb);
            S.tsched_dev.alloc(1);
            IRH_CHECK(hipMemcpyAsync(S.tsched_dev.p, &S.tsched, sizeof(BcrTopSched), hipMemcpyHostToDevice, g.stream));
            IRH_CHECK(hipStreamSynchronize(g.stream));
            break;
        }
        if (l == 0 && nb <= 8) S.top16 = false;
        L.W.alloc((size_t)L.nch * 7 * B * NC);
        if (l > 0) L.x.alloc((size_t)L.nch * 8 * B * NR);
        if (nb <= 8 && !g.bcr_shard) {
            S.top16 = false;
            break;
        }
        L.sepD.alloc((size_t)L.nch * B * B);
        L.extD.alloc((size_t)L.nch * B * B);
        L.extG.alloc((size_t)L.nch * B * B);
        L.sepR.alloc((size_t)L.nch * B * NR);
        L.extR.alloc((size_t)L.nch * B * NR);
        if (nb <= 8) break;
        nb = L.nch + (l == 0 ? nraw : 0);
        nch = (nb + 7) / 8;
    }
    if (g.bcr_shard) {
        S.ext0 = g.bcr_ext0;
        S.ghost_extcol.upload(g.bcr_ghost_extcol, g.stream);
        S.remD.alloc((size_t)B * B);
        S.remR.alloc((size_t)B * NR);
        IRH_CHECK(hipStreamSynchronize(g.stream));
    }
    S.xtop.alloc((size_t)B * NR);
    if (S.nfar > 0) bcr_closure_plan(g);
    if (g.bcr_shard && !g.bcr_fix_row.empty()) {
        S.nfix = (int)g.bcr_fix_row.size();
        S.fix_row.upload(g.bcr_fix_row, g.stream);
        S.fix_off.upload(g.bcr_fix_off, g.stream);
        S.fix_e.upload(g.bcr_fix_e, g.stream);
        S.fix_saved.alloc((size_t)S.nfix);
        IRH_CHECK(hipStreamSynchronize(g.stream));
    }
}

// open_top: the last level writes its separator data like every other level instead of solving it (a shard: the
// separators of all ranks form the top system, bcr_top_solve); xc_top / xext: its solution and the solution of the
// previous rank's separator for the way back
template <int B, int NR>
static void bcr_run(Graph &g, int only, int pass, bool open_top = false, int phase = 0, const double *xc_top = nullptr,
                    const double *xext = nullptr) {
    BcrState &S = *g.bcr;
    Level &L0 = g.levels[0];
    hipStream_t st = g.stream;
    const int nl = (int)S.lev.size();
    // (the stamp helpers steer ONE handle through BcrState::dbg_word instead of rewriting the process environment under
    // other handles' and threads' solves -- advisor, round 5; the library itself never writes the environment)
    const char *env_s = getenv("IROTAVG_BCR_DBG");
    const int env_dbg = env_s ? atoi(env_s) : 0;
    const int dbg = S.dbg_word >= 0 ? S.dbg_word : env_dbg;
    long long *stamps = nullptr;
    if (dbg & 64) {
        if (!S.stamps.p) S.stamps.alloc((size_t)S.lev[0].nch * 16);
        stamps = S.stamps.p;
    }
    const int nfar = 0;
    const int *fi = nullptr, *fj = nullptr;
    (void)pass;
    // the reductions of all levels above level 0 in one launch (k_bcr_reduce_up): three right-hand sides, blocks up to 24
    // (the eight-wave workgroups), at least two such levels, level 1 of at most 256 chunks (one workgroup per CU: all
    // resident), no closures (their inverses and dead-pivot count ride on the separate launches), not a shard, not
    // when single levels are timed
    bool fused_up = false;
    if constexpr (NR == 3 && B <= 24) {
        fused_up = nl >= 3 && nl - 1 <= kUpLevels && only < 0 && phase == 0 && !open_top && !g.bcr_shard && S.nfar == 0 &&
                   S.lev[1].nch <= 256 && !(dbg & 64) && !getenv("IROTAVG_BCR_NARROW") && !getenv("IROTAVG_BCR_NO_FUSED_UP") &&
                   !g.bcr_no_fused_up;
        if ((dbg & 128) && !S.stamps.p) S.stamps.alloc((size_t)std::max(S.lev[0].nch * 16, 32 * kUpLevels));
        if (fused_up) fused_up = bcr_up_reserve<B>(g, S.lev[1].nch);
    }
    if (only < 0 && phase != 2) S.up_used = false;
    for (int l = 0; l < nl && phase != 2; l++) {
        if (only >= 0 && only != l) continue;
        if constexpr (NR == 3 && B <= 24) {
            if (S.top16 && l == nl - 1 && !fused_up) {
                // the single-workgroup top in a launch of its own
                const BcrLevel &F = S.lev[l - 1];
                const size_t lds = bcr_top_lds(B, S.lev[l].nb);
                static std::atomic<size_t> lds_set[16];
                if (lds_set[g.device & 15].load() < lds) {
                    IRH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bcr_top<B>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    lds_set[g.device & 15].store(lds);
                }
                hipLaunchKernelGGL((k_bcr_top<B>), dim3(1), dim3(512), lds, st, S.tsched_dev.p, F.sepD.p, F.sepR.p, F.extD.p,
                                   F.extR.p, F.extG.p, S.lev[l].x.p, dbg, stamps);
                continue;
            }
            if (fused_up && l >= 1) {
                if (l > 1) continue;
                if (S.up_cnt.n < (size_t)kUpLevels + 1) {
                    S.up_cnt.alloc(kUpLevels + 1);
                    S.up_cnt.zero(st);
                    S.up_gen = 0;
                }
                if (!S.up_args.p) {
                    // the launch's arguments: written once per handle
                    BcrUpArgs A;
                    memset(&A, 0, sizeof(A));
                    A.nl = nl - 1;
                    for (int i = 0; i < A.nl; i++) {
                        BcrLevel &Li = S.lev[1 + i], &Fi = S.lev[i];
                        A.lev[i] = BcrUpLevel{Li.nb,          Li.nred,        Li.nch,         (gp_cd)Fi.sepD.p, (gp_cd)Fi.sepR.p,
                                              (gp_cd)Fi.extD.p, (gp_cd)Fi.extR.p, (gp_cd)Fi.extG.p, (gp_d)Li.W.p,   (gp_d)Li.sepD.p,
                                              (gp_d)Li.sepR.p, (gp_d)Li.extD.p, (gp_d)Li.extR.p, (gp_d)Li.extG.p};
                    }
                    A.n = L0.n;
                    A.sl_off = (gp_ci)L0.sl_off.p;
                    A.col = (gp_ci)L0.col.p;
                    A.val = (gp_cd)L0.val.p;
                    A.diag = (gp_cd)L0.diag.p;
                    A.rhs = (gp_cd4)L0.b.p;
                    A.xtop = (gp_d)S.xtop.p;
                    A.cnt = (gp_u)S.up_cnt.p;
                    A.top16 = S.top16;
                    A.top = S.tsched;
                    A.xtop_all = (gp_d)S.lev[nl - 1].x.p;
                    A.fail = (gp_i) reinterpret_cast<int *>(S.up_cnt.p + kUpLevels);
                    S.up_args.alloc(1);
                    IRH_CHECK(hipMemcpyAsync(S.up_args.p, &A, sizeof(A), hipMemcpyHostToDevice, st));
                    IRH_CHECK(hipStreamSynchronize(st));  // (A is on this stack)
                }
                // (tests: the give-up path without a second process -- the word is set as if a wait had timed out)
                if (getenv("IROTAVG_BCR_FAKE_UP_FAIL") && g.stats.direct_up_fallbacks == 0)
                    IRH_CHECK(hipMemsetAsync(S.up_cnt.p + kUpLevels, 1, sizeof(int), st));
                const size_t lds = bcr_up_lds<B>(S.top16 ? S.lev[nl - 1].nb : 0);
                static std::atomic<size_t> lds_set[16];
                if (lds_set[g.device & 15].load() < lds) {
                    IRH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bcr_reduce_up<B>),
                                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                    lds_set[g.device & 15].store(lds);
                }
                hipLaunchKernelGGL((k_bcr_reduce_up<B>), dim3(S.lev[1].nch), dim3(512), lds, st, S.up_args.p, ++S.up_gen, dbg,
                                   (dbg & 128) ? S.stamps.p : (long long *)nullptr);
                S.up_used = true;
                continue;
            }
        }
        BcrLevel &L = S.lev[l];
        const BcrLevel *F = l > 0 ? &S.lev[l - 1] : nullptr;
        const bool top = l == nl - 1 && !open_top;
#define IRH_BCR_ARGS                                                                                             \
    L.nb, L.nred, L0.n, L0.sl_off.p, L0.col.p, L0.val.p, L0.diag.p, L0.b.p, F ? F->sepD.p : nullptr,             \
        F ? F->sepR.p : nullptr, F ? F->extD.p : nullptr, F ? F->extR.p : nullptr, F ? F->extG.p : nullptr, L.W.p, \
        L.sepD.p, L.sepR.p, L.extD.p, L.extR.p, L.extG.p, S.xtop.p, dbg, nfar, fi, fj, S.ext0, g.bptr.p, g.bghost.p,  \
        g.bval.p, S.ghost_extcol.p, (int)g.bcr_shard, stamps, S.nfar > 0 ? S.Dinv[(size_t)l].p : (double *)nullptr,         \
        S.nfar > 0 ? S.topDinv.p : (double *)nullptr, S.nfar > 0 ? S.dead.p : (int *)nullptr, (int)g.bcr_guard
        // eight waves per chunk when every chunk has a CU to itself (see k_bcr_reduce)
        const bool wide = L.nch <= 256 && !getenv("IROTAVG_BCR_NARROW");
#define IRH_BCR_LAUNCH(L0_, TOP_)                                                                                   \
    if constexpr (B <= 24) {                                                                                        \
        if (wide)                                                                                                   \
            hipLaunchKernelGGL((k_bcr_reduce<B, NR, L0_, TOP_, 8>), dim3(L.nch), dim3(512), 0, st, IRH_BCR_ARGS);   \
        else                                                                                                        \
            hipLaunchKernelGGL((k_bcr_reduce<B, NR, L0_, TOP_, 4>), dim3(L.nch), dim3(256), 0, st, IRH_BCR_ARGS);   \
    } else {                                                                                                        \
        hipLaunchKernelGGL((k_bcr_reduce<B, NR, L0_, TOP_, 4>), dim3(L.nch), dim3(256), 0, st, IRH_BCR_ARGS);       \
    }
        if (l == 0 && top) {
            IRH_BCR_LAUNCH(true, true)
        } else if (l == 0) {
            IRH_BCR_LAUNCH(true, false)
        } else if (top) {
            IRH_BCR_LAUNCH(false, true)
        } else {
            IRH_BCR_LAUNCH(false, false)
        }
#undef IRH_BCR_LAUNCH
#undef IRH_BCR_ARGS
    }
    // the ways back of the levels above level 0 in one launch (k_bcr_back_top) unless this is a shard, the closures'
    // columns ride along or IROTAVG_BCR_NO_FUSED_BACK is set
    double4 *Xout = g.bcr_out ? g.bcr_out : g.X.p + g.ng;
    bool fused_back = false;
    // (a single-workgroup top has made its own way back: the levels below it are left)
    const int ltop = S.top16 ? nl - 2 : nl - 1;
    if constexpr (NR == 3) fused_back = ltop >= 1 && !g.bcr_shard && !open_top && !getenv("IROTAVG_BCR_NO_FUSED_BACK");
    // K6 inside the ways back (run_irls asked for it and bcr_apply_ok() said yes): every solution row of level 0 is written
    // by k_bcr_back (level 0's chunks: slots 0 .. nch0 - 1) or, on a mixed level 1, by k_bcr_back_top (slots nch0 ...)
    const bool apply = NR == 3 && g.bcr_apply && only < 0 && phase == 0 && (fused_back || (S.top16 && nl == 2)) && !g.bcr_out &&
                       g.ng == 0;
    g.bcr_applied = apply;
    for (int l = ltop; l >= 0 && phase != 1; l--) {
        if (only >= 0 && only != 100 + l) continue;
        BcrLevel &L = S.lev[l];
        if constexpr (NR == 3) {
            if (fused_back && l >= 1) {
                if (l > 1) continue;
                BcrBackPlan P;
                for (int k = 0; k < nl; k++) {
                    P.W[k] = S.lev[k].W.p;
                    P.nb[k] = S.lev[k].nb;
                }
                P.top = ltop;
                P.base = 1;
                hipLaunchKernelGGL((k_bcr_back_top<B>), dim3(L.nch), dim3(kBackTopThreads), 0, st, P,
                                   S.top16 ? S.lev[nl - 1].x.p : S.xtop.p, L.x.p, Xout, L0.n, L.nred,
                                   apply ? g.Q.p : (double4 *)nullptr, g.f, g.part_score.p, S.lev[0].nch, (int)S.top16);
                continue;
            }
        }
        const double *xc = l == nl - 1 ? (open_top ? xc_top : S.xtop.p) : S.lev[l + 1].x.p;
        if (l == 0)
            hipLaunchKernelGGL((k_bcr_back<B, NR, true>), dim3(L.nch), dim3(256), 0, st, L.nb, L.nred, L0.n, L.W.p, xc,
                               (double *)nullptr, Xout, (double *)nullptr, 0, 0, nfar, xext, (int)g.bcr_shard,
                               apply ? g.Q.p : (double4 *)nullptr, g.f, g.part_score.p, 0);
        else
            hipLaunchKernelGGL((k_bcr_back<B, NR, false>), dim3(L.nch), dim3(256), 0, st, L.nb, L.nred, L0.n, L.W.p, xc,
                               L.x.p, Xout, (double *)nullptr, 0, 0, nfar, xext, (int)g.bcr_shard);
    }
}

template <int B>
static void bcr_launch_forward(hipStream_t st, const BcrClPlan &P, int r, int nslots, const int *off, const int4 *steps,
                               const int4 *init, double *recR, double *recW, double *T, const double *slot_init,
                               const int2 *fin, const int *gid, double *dep, int world, int rank);

template <int B>
static void bcr_run_all(Graph &g, int only) {
    BcrState &S = *g.bcr;
    if (S.nfar == 0 || only >= 0) {
        bcr_run<B, 3>(g, only, 0);
        return;
    }
    // closures: the reduction (which keeps the inverses of the eliminated blocks), the closures' forward eliminations,
    // the Woodbury system, the corrected right-hand sides, the ways back
    hipStream_t st = g.stream;
    const int r = S.nfar, npad = S.cl_npad, nl = (int)S.lev.size();
    if (!S.dead_clean) IRH_CHECK(hipMemsetAsync(S.dead.p, 0, sizeof(int), st));
    S.dead_clean = false;
    bcr_run<B, 3>(g, -1, 0, false, 1);
    BcrClPlan P;
    for (int l = 0; l < nl; l++) {
        P.W[l] = S.lev[l].W.p;
        P.Wrw[l] = S.lev[l].W.p;
        P.Dinv[l] = S.Dinv[(size_t)l].p;
    }
    P.topDinv = S.topDinv.p;
    P.xtop = S.xtop.p;
    bcr_launch_forward<B>(st, P, r, S.cl_nslots, S.cl_off.p, S.cl_step.p, S.cl_init.p, S.cl_R.p, S.cl_W.p, S.cl_T.p, nullptr,
                          nullptr, nullptr, nullptr, 0, 0);
    const int nt = (npad + 15) / 16;
    static const bool s_tiles = getenv("IROTAVG_BCR_S_TILES") != nullptr;  // A/B: the tile kernel for every closure count
    const size_t lds_pairs = (size_t)4 * (S.cl_maxsteps + 64) * sizeof(int);
    if (r <= kClosurePairsMax && S.cl_maxsteps < 32768 && lds_pairs <= 60 * 1024 && !s_tiles)
        hipLaunchKernelGGL((k_bcr_closure_S_pairs<B>), dim3((npad + 3) / 4, npad), dim3(256), lds_pairs, st, r, npad,
                           S.cl_maxsteps, S.cl_off.p, S.cl_step.p, S.cl_R.p, S.cl_W.p, S.far_e.p, g.bcr_wsrc, g.bcr_wsquare,
                           S.cl_S.p, S.cl_T.p, S.cl_alive.p);
    else
    hipLaunchKernelGGL((k_bcr_closure_S<B>), dim3(nt, nt), dim3(256), (size_t)32 * S.cl_maxsteps * sizeof(int), st, r, npad,
                       S.cl_maxsteps, S.cl_ndense, S.cl_dense.p, S.cl_off.p, S.cl_step.p, S.cl_R.p,
                       S.cl_W.p, S.far_e.p, g.bcr_wsrc, g.bcr_wsquare, S.cl_S.p, S.cl_T.p, S.cl_alive.p);
    if (r <= kSolveLdsMax) {
        bcr_launch_solve_lds(st, r, npad, S.cl_S.p, S.cl_T.p, S.lam.p);
    } else {
        dense_invert_spd(g, S.cl_S.p, npad);
        hipLaunchKernelGGL(k_bcr_closure_lambda, dim3((r + 3) / 4), dim3(256), 0, st, r, npad, S.cl_S.p, S.cl_T.p,
                           S.lam.p);
    }
    hipLaunchKernelGGL((k_bcr_closure_correct<B>), dim3(S.cl_nelim), dim3(1024), 0, st, P, S.co_off.p, S.co_rec.p,
                       S.co_elim.p, S.cl_owner.p, S.cl_W.p, S.lam.p);
    bcr_run<B, 3>(g, -1, 0, false, 2);
}

// ---------------------------------------------------------------------------------------------
// the sharded form (dist.hip): a rank's range of the sequence is reduced to its last block; the separators of the
// `world` <= 8 ranks are one chunk, solved redundantly by every rank after ONE gather
// ---------------------------------------------------------------------------------------------
struct BcrPtrs {
    const double *d[kMaxLevels], *r[kMaxLevels];
    int n;
};
// remote contributions summed over the shard's levels (chunk 0 of each) and, with the top level's separator data,
// written into this rank's slices of the top buffer
template <int B>
__global__ __launch_bounds__(256) void k_bcr_pack_top(BcrPtrs P, int ext0, const double *__restrict__ sepD,
                                                      const double *__restrict__ sepR, const double *__restrict__ extG,
                                                      double *__restrict__ buf, int world, int rank) {
    constexpr int BB = B * B, BR = B * 3;
    double *oD = buf + (size_t)rank * BB, *oXD = buf + (size_t)world * BB + (size_t)rank * BB,
           *oXG = buf + (size_t)2 * world * BB + (size_t)rank * BB,
           *oR = buf + (size_t)3 * world * BB + (size_t)rank * BR,
           *oXR = buf + (size_t)3 * world * BB + (size_t)world * BR + (size_t)rank * BR;
    for (int e = threadIdx.x; e < BB; e += 256) {
        oD[e] = sepD[e];
        double sacc = 0.0;
        if (ext0)
            for (int l = 0; l < P.n; l++) sacc += P.d[l][e];
        oXD[e] = sacc;
        oXG[e] = ext0 ? extG[e] : 0.0;
    }
    for (int e = threadIdx.x; e < BR; e += 256) {
        oR[e] = sepR[e];
        double sacc = 0.0;
        if (ext0)
            for (int l = 0; l < P.n; l++) sacc += P.r[l][e];
        oXR[e] = sacc;
    }
}

// rec (rank-major records of one rank's five slices, what ncclAllGather moves) <-> buf (array-major, what the separator
// system is read from): mine >= 0: buf's slices of rank `mine` -> its record; mine < 0: every record -> buf
__global__ __launch_bounds__(256) void k_bcr_top_records(int B, int world, int mine, double *__restrict__ buf, double *__restrict__ rec) {
    const int BB = B * B, BR = B * 3, RS = 3 * BB + 2 * BR;
    const int r0 = mine >= 0 ? mine : 0, r1 = mine >= 0 ? mine + 1 : world;
    for (int idx = blockIdx.x * 256 + threadIdx.x; idx < (r1 - r0) * RS; idx += gridDim.x * 256) {
        const int rank = r0 + idx / RS, e = idx % RS;
        size_t b;   // position in buf
        if (e < 3 * BB) b = (size_t)(e / BB) * world * BB + (size_t)rank * BB + e % BB;
        else {
            const int q = e - 3 * BB;
            b = (size_t)3 * world * BB + (size_t)(q / BR) * world * BR + (size_t)rank * BR + q % BR;
        }
        if (mine >= 0) rec[(size_t)rank * RS + e] = buf[b];
        else buf[b] = rec[(size_t)rank * RS + e];
    }
}
void bcr_top_to_record(BcrTop &T, int rank, hipStream_t st) {
    hipLaunchKernelGGL(k_bcr_top_records, dim3(8), dim3(256), 0, st, T.B, T.world, rank, T.buf.p, T.rec.p);
}
void bcr_top_from_records(BcrTop &T, hipStream_t st) {
    hipLaunchKernelGGL(k_bcr_top_records, dim3(8 * T.world), dim3(256), 0, st, T.B, T.world, -1, T.buf.p, T.rec.p);
}

void bcr_top_alloc(BcrTop &T, int B, int world) {
    T.B = B;
    T.world = world;
    T.buf.alloc(T.n_doubles());
    T.rec.alloc(T.n_doubles());
    T.W.alloc((size_t)7 * B * (2 * B + 3));
    T.x.alloc((size_t)8 * B * 3);
    T.xtop.alloc((size_t)B * 3);
}

// ---- loop closures on a sharded sequence (round 5) -------------------------------------------------------------
// The Woodbury correction of "The closures' part of a solve" with the elimination tree cut at the ranks' separators:
// a closure's incidence column is eliminated through the levels of the rank(s) that hold its endpoints
// (k_bcr_closure_forward on the shard's own step programs), what it leaves on the rank's separator and on the one
// before it is the column's right-hand side of the separator system, which every rank eliminates for ALL closures
// after the ranks' buffers have been summed; S = C^-1 + V' A_b^-1 V and T = V' Y are sums over eliminated blocks --
// a rank's blocks (its share, added to the same buffer) and the separator system's (added by everybody after the sum);
// lambda is solved by every rank, the corrections of the stored right-hand sides are local again.

// the rows with a closure to a GHOST view: their diagonal holds that closure's weight (a ghost edge is Dirichlet mass
// on a shard) -- the band operator loses it for the time of the reduction as bcr_gather_row takes a local far entry out
__global__ __launch_bounds__(256) void k_bcr_fix_diag(int nfix, const int *__restrict__ row, const int *__restrict__ off,
                                                       const int *__restrict__ e, const double *__restrict__ wsrc, int wsquare,
                                                       double *__restrict__ diag, double *__restrict__ saved, int restore) {
    const int k = blockIdx.x * 256 + threadIdx.x;
    if (k >= nfix) return;
    const int rw = row[k];
    if (restore) {
        diag[rw] = saved[k];
        return;
    }
    const double d = diag[rw];
    saved[k] = d;
    double s = 0.0;
    for (int t = off[k]; t < off[k + 1]; t++) {
        const double w = wsrc[e[t]];
        s += wsquare ? w * w : w;
    }
    diag[rw] = d - s;
}

__global__ void k_bcr_add_dead(const int *__restrict__ dead, double *__restrict__ sum) {
    if (threadIdx.x == 0 && blockIdx.x == 0) *sum += (double)*dead;
}

// After the sum over the ranks: S += the separator system's share (sum over its eliminations of R_p . D^-1 R_q: every
// closure has a record on every one of them -- a product of two r x (nst B) arrays), T += its share; a closure its owner
// flagged as dead (weight 0) gets the identity's row and column and T = 0; the padding of the inversion.
template <int B>
__global__ __launch_bounds__(256) void k_bcr_top_closure_S(int r, int npad, int nst, const double *__restrict__ recR,
                                                            const double *__restrict__ recW, const double *__restrict__ Ttop,
                                                            double *__restrict__ S, double *__restrict__ T,
                                                            const double *__restrict__ deadx) {
    __shared__ double sx[16][65], sy[16][65];
    if (blockIdx.x < blockIdx.y) return;
    const int tp = threadIdx.x >> 4, tq = threadIdx.x & 15;
    const int p = blockIdx.y * 16 + tp, q = blockIdx.x * 16 + tq;
    const int K = nst * B;
    double sum = 0.0;
    for (int k0 = 0; k0 < K; k0 += 64) {
        for (int e = threadIdx.x; e < 32 * 64; e += 256) {
            const int who = e >> 6, k = e & 63;
            const int cl = who < 16 ? blockIdx.y * 16 + who : blockIdx.x * 16 + who - 16;
            double v = 0.0;
            if (cl < r && k0 + k < K) v = (who < 16 ? recR : recW)[(size_t)cl * K + k0 + k];
            if (who < 16)
                sx[who][k] = v;
            else
                sy[who - 16][k] = v;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < 64; k++) sum = fma(sx[tp][k], sy[tq][k], sum);
        __syncthreads();
    }
    if (p >= npad || q >= npad || q < p) return;
    if (p >= r || q >= r) {
        S[(size_t)p * npad + q] = S[(size_t)q * npad + p] = p == q ? 1.0 : 0.0;
        return;
    }
    const bool live = !(deadx[p] > 0.0) && !(deadx[q] > 0.0);
    const double v = live ? S[(size_t)p * npad + q] + sum : (p == q ? 1.0 : 0.0);
    S[(size_t)p * npad + q] = v;
    S[(size_t)q * npad + p] = v;
    if (p == q)
        for (int c = 0; c < 3; c++) T[3 * p + c] = live ? T[3 * p + c] + Ttop[3 * p + c] : 0.0;
}

template <int B>
static BcrClPlan bcr_local_clplan(BcrState &S) {
    BcrClPlan P;
    const int nl = (int)S.lev.size();
    for (int l = 0; l < kMaxLevels; l++) {
        P.W[l] = P.Wrw[l] = l < nl ? S.lev[l].W.p : nullptr;
        P.Dinv[l] = l < nl && l < (int)S.Dinv.size() ? S.Dinv[(size_t)l].p : nullptr;
    }
    P.topDinv = S.topDinv.p;
    P.xtop = S.xtop.p;
    return P;
}

template <int B>
static void bcr_launch_forward(hipStream_t st, const BcrClPlan &P, int r, int nslots, const int *off, const int4 *steps,
                               const int4 *init, double *recR, double *recW, double *T, const double *slot_init,
                               const int2 *fin, const int *gid, double *dep, int world, int rank) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    // Few closures: one workgroup per closure, four staged blocks of B x (3B + 3) doubles (58 KB at B = 24, 100 KB at
    // B = 32) + the slots -- the chain of a closure's steps is what takes the time, and two such workgroups fill a CU.
    // Many closures (more than two per CU): a wave per closure, LDS for the slots only (measured at a thousand closures:
    // 65 us against 100).
    const size_t lds_wg = ((size_t)4 * B * (3 * B + 3) + (size_t)4 * (3 * B + 3) + (size_t)nslots * B) * sizeof(double);
    // (blocks of 32: 100 KB, one workgroup per CU -- the chip takes 256 closures at a time then)
    const int r_wg = lds_wg * 2 <= 150 * 1024 ? 400 : 256;
    // EIGHT steps per round trip (eight waves, twice the stages: IROTAVG_BCR_FORWARD8=1) was built and measured in round 6:
    // 50 us against 35.5 at thirty closures -- the steps are bound by their two barriers each (sixteen waves' worth with
    // 512 threads), not by the round trips. Kept behind the switch as the record of that measurement.
    constexpr int kForward8Max = 128;
    const size_t lds_wg8 = ((size_t)8 * B * (3 * B + 3) + (size_t)4 * (3 * B + 3) + (size_t)nslots * B) * sizeof(double);
    if (r <= kForward8Max && lds_wg8 <= 150 * 1024 && !getenv("IROTAVG_BCR_FORWARD_WAVE") && getenv("IROTAVG_BCR_FORWARD8")) {
        static std::atomic<size_t> lds_set8[16];
        if (lds_wg8 > 64 * 1024 && lds_set8[dev & 15].load() < lds_wg8) {
            IRH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bcr_closure_forward<B, 8>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_wg8));
            lds_set8[dev & 15].store(lds_wg8);
        }
        hipLaunchKernelGGL((k_bcr_closure_forward<B, 8>), dim3(r), dim3(512), lds_wg8, st, P, r, nslots, off, steps, init, recR,
                           recW, T, slot_init, fin, gid, dep, world, rank);
        return;
    }
    if (r <= r_wg && lds_wg <= 150 * 1024 && !getenv("IROTAVG_BCR_FORWARD_WAVE")) {
        static std::atomic<size_t> lds_set[16];
        if (lds_wg > 64 * 1024 && lds_set[dev & 15].load() < lds_wg) {
            IRH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bcr_closure_forward<B>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_wg));
            lds_set[dev & 15].store(lds_wg);
        }
        hipLaunchKernelGGL((k_bcr_closure_forward<B>), dim3(r), dim3(256), lds_wg, st, P, r, nslots, off, steps, init, recR, recW,
                           T, slot_init, fin, gid, dep, world, rank);
        return;
    }
    int wpb = 4;
    while (wpb > 1 && (size_t)wpb * nslots * B * sizeof(double) > 60 * 1024) wpb >>= 1;
    const size_t lds = (size_t)wpb * nslots * B * sizeof(double);
    if (lds > 64 * 1024)
        IRH_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bcr_closure_forward_wave<B>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL((k_bcr_closure_forward_wave<B>), dim3((r + wpb - 1) / wpb), dim3(64 * wpb), lds, st, P, r, nslots, off,
                       steps, init, recR, recW, T, slot_init, fin, gid, dep, world, rank);
}

template <int B>
static void bcr_shard_reduce_t(Graph &g, BcrTop &T, int rank) {
    BcrState &S = *g.bcr;
    if (S.nfar > 0) IRH_CHECK(hipMemsetAsync(S.dead.p, 0, sizeof(int), g.stream));
    if (S.nfix > 0)
        hipLaunchKernelGGL(k_bcr_fix_diag, dim3((S.nfix + 255) / 256), dim3(256), 0, g.stream, S.nfix, S.fix_row.p, S.fix_off.p,
                           S.fix_e.p, g.bcr_wsrc, g.bcr_wsquare, g.levels[0].diag.p, S.fix_saved.p, 0);
    bcr_run<B, 3>(g, -1, 0, true, 1);
    if (S.nfix > 0)
        hipLaunchKernelGGL(k_bcr_fix_diag, dim3((S.nfix + 255) / 256), dim3(256), 0, g.stream, S.nfix, S.fix_row.p, S.fix_off.p,
                           S.fix_e.p, g.bcr_wsrc, g.bcr_wsquare, g.levels[0].diag.p, S.fix_saved.p, 1);
    BcrPtrs P;
    P.n = (int)S.lev.size();
    for (int l = 0; l < P.n; l++) {
        P.d[l] = S.lev[l].extD.p;  // chunk 0 of every level
        P.r[l] = S.lev[l].extR.p;
    }
    BcrLevel &L = S.lev.back();
    hipLaunchKernelGGL((k_bcr_pack_top<B>), dim3(1), dim3(256), 0, g.stream, P, S.ext0, L.sepD.p, L.sepR.p, L.extG.p,
                       T.buf.p, T.world, rank);
}
template <int B>
static void bcr_top_reduce_launch(Graph &g, BcrTop &T, double *Dinv, double *topDinv, int *dead, int reg = 0) {
    const int W = T.world;
    const size_t BB = (size_t)B * B, BR = (size_t)B * 3;
    double *buf = T.buf.p;
    const double *nul = nullptr;
    hipLaunchKernelGGL((k_bcr_reduce<B, 3, false, true, 4>), dim3(1), dim3(256), 0, g.stream, W, W, 0, (const int *)nullptr,
                       (const int *)nullptr, nul, nul, (const double4 *)nullptr, buf, buf + 3 * W * BB, buf + W * BB,
                       buf + 3 * W * BB + W * BR, buf + 2 * W * BB, T.W.p, (double *)nullptr, (double *)nullptr,
                       (double *)nullptr, (double *)nullptr, (double *)nullptr, T.xtop.p, 0, 0, (const int *)nullptr,
                       (const int *)nullptr, 0, (const int *)nullptr, (const int *)nullptr, nul, (const int *)nullptr, 0,
                       (long long *)nullptr, Dinv, topDinv, dead, reg);
}
template <int B>
static void bcr_top_back_launch(Graph &g, BcrTop &T) {
    const int W = T.world;
    const double *nul = nullptr;
    hipLaunchKernelGGL((k_bcr_back<B, 3, false>), dim3(1), dim3(256), 0, g.stream, W, W, 0, T.W.p, T.xtop.p, T.x.p,
                       (double4 *)nullptr, (double *)nullptr, 0, 0, 0, nul, 0);
}
template <int B>
static void bcr_top_solve_t(Graph &g, BcrTop &T) {
    bcr_top_reduce_launch<B>(g, T, nullptr, nullptr, nullptr);
    bcr_top_back_launch<B>(g, T);
}
template <int B>
static void bcr_shard_back_t(Graph &g, BcrTop &T, int rank) {
    const double *xs = T.x.p + (size_t)rank * B * 3;
    bcr_run<B, 3>(g, -1, 0, true, 2, xs, rank > 0 ? xs - (size_t)B * 3 : nullptr);
}

// this rank's closures: their forward eliminations through its levels, its share of S and T and its dead pivots -> T.xbuf
template <int B>
static void bcr_shard_closures_forward_t(Graph &g, BcrTop &T, int rank) {
    BcrState &S = *g.bcr;
    if (S.nfar == 0) return;
    hipStream_t st = g.stream;
    const BcrClPlan P = bcr_local_clplan<B>(S);
    const int r = S.nfar;
    bcr_launch_forward<B>(st, P, r, S.cl_nslots, S.cl_off.p, S.cl_step.p, S.cl_init.p, S.cl_R.p, S.cl_W.p, S.cl_T.p, nullptr,
                          S.cl_fin.p, S.cl_gid.p, T.xbuf.p + T.x_dep(), T.world, rank);
    const int nt = (S.cl_npad + 15) / 16;
    hipLaunchKernelGGL((k_bcr_closure_S<B>), dim3(nt, nt), dim3(256), (size_t)32 * S.cl_maxsteps * sizeof(int), st, r,
                       S.cl_npad, S.cl_maxsteps, S.cl_ndense, S.cl_dense.p, S.cl_off.p, S.cl_step.p, S.cl_R.p, S.cl_W.p,
                       S.far_e.p, g.bcr_wsrc, g.bcr_wsquare, T.xbuf.p + T.x_S(), S.cl_T.p, S.cl_alive.p, S.cl_gid.p,
                       S.cl_own.p, T.npad, T.xbuf.p + T.x_T(), T.xbuf.p + T.x_dead());
    hipLaunchKernelGGL(k_bcr_add_dead, dim3(1), dim3(64), 0, st, S.dead.p, T.xbuf.p + T.x_pivots());
}

// the separator system with the closures' columns (every rank, after the sum of the ranks' buffers): factor, the columns'
// forward eliminations, S and T completed, lambda, the corrected right-hand sides, the separators' solution
template <int B>
static void bcr_top_solve_closures_t(Graph &g, BcrTop &T) {
    hipStream_t st = g.stream;
    const int r = T.r, npad = T.npad, W = T.world;
    IRH_CHECK(hipMemsetAsync(T.dead.p, 0, sizeof(int), st));
    bcr_top_reduce_launch<B>(g, T, T.Dinv.p, T.topDinv.p, T.dead.p, (int)g.bcr_guard);
    BcrClPlan P;
    for (int l = 0; l < kMaxLevels; l++) {
        P.W[l] = P.Wrw[l] = l == 0 ? T.W.p : nullptr;
        P.Dinv[l] = l == 0 ? T.Dinv.p : nullptr;
    }
    P.topDinv = T.topDinv.p;
    P.xtop = T.xtop.p;
    double *xb = T.xbuf.p;
    bcr_launch_forward<B>(st, P, r, W, T.cl_off.p, T.cl_step.p, nullptr, T.recR.p, T.recW.p, T.Ttop.p, xb + T.x_dep(), nullptr,
                          nullptr, nullptr, 0, 0);
    const int nt = (npad + 15) / 16;
    hipLaunchKernelGGL((k_bcr_top_closure_S<B>), dim3(nt, nt), dim3(256), 0, st, r, npad, T.nst, T.recR.p, T.recW.p, T.Ttop.p,
                       xb + T.x_S(), xb + T.x_T(), xb + T.x_dead());
    if (r <= kSolveLdsMax) {
        bcr_launch_solve_lds(st, r, npad, xb + T.x_S(), xb + T.x_T(), T.lam.p);
    } else {
        dense_invert_spd(g, xb + T.x_S(), npad);
        hipLaunchKernelGGL(k_bcr_closure_lambda, dim3((r + 3) / 4), dim3(256), 0, st, r, npad, xb + T.x_S(), xb + T.x_T(),
                           T.lam.p);
    }
    hipLaunchKernelGGL((k_bcr_closure_correct<B>), dim3(T.nst), dim3(1024), 0, st, P, T.co_off.p, T.co_rec.p, T.co_elim.p,
                       T.cl_owner.p, T.recW.p, T.lam.p);
    bcr_top_back_launch<B>(g, T);
}

// the stored right-hand sides of this rank's blocks on the closures' paths -= sum_q (D^-1 R_q) lambda_q'
template <int B>
static void bcr_shard_closures_correct_t(Graph &g, BcrTop &T) {
    BcrState &S = *g.bcr;
    if (S.nfar == 0 || S.cl_nelim == 0) return;
    const BcrClPlan P = bcr_local_clplan<B>(S);
    hipLaunchKernelGGL((k_bcr_closure_correct<B>), dim3(S.cl_nelim), dim3(1024), 0, g.stream, P, S.co_off.p, S.co_rec.p,
                       S.co_elim.p, S.cl_owner.p, S.cl_W.p, T.lam.p);
}

// the separator system's step program: one chunk of `world` blocks at their own positions, nothing before it; every
// closure runs every elimination (slot t = block t, started from the summed deposits); the eighth block is solved
void bcr_top_closures_alloc(Graph &g0, BcrTop &T, int r) {
    T.r = r;
    if (r <= 0) return;
    const int B = T.B, W = T.world;
    T.npad = r <= kSolveLdsMax ? r : (r + 63) / 64 * 64;
    static const int sched[7] = {0, 2, 4, 6, 1, 5, 3};
    std::vector<int4> prog;
    std::vector<int2> elim;
    for (int k = 0; k < 7; k++) {
        const int i = sched[k];
        if (i >= W) continue;
        int a, c;
        if (k < 4) {
            a = i - 1;
            c = i + 1;
        } else if (k < 6) {
            a = i == 1 ? -1 : 3;
            c = i == 1 ? 3 : 7;
        } else {
            a = -1;
            c = 7;
        }
        const int dA = a >= 0 ? a : 255, dC = c < W ? c : 255;
        prog.push_back(int4{0, i, i | (dA << 8) | (dC << 16), 0});
        elim.push_back(int2{0, i});
    }
    if (W == 8) {
        prog.push_back(int4{-1, 0, 7 | (255 << 8) | (255 << 16), 0});
        elim.push_back(int2{-1, 0});
    }
    const int nst = (int)prog.size();
    T.nst = nst;
    std::vector<int> off((size_t)r + 1), owner((size_t)r * nst), eoff((size_t)nst + 1), erec((size_t)r * nst);
    std::vector<int4> steps((size_t)r * nst);
    for (int q = 0; q <= r; q++) off[(size_t)q] = q * nst;
    for (int q = 0; q < r; q++)
        for (int d = 0; d < nst; d++) {
            steps[(size_t)q * nst + d] = prog[(size_t)d];
            owner[(size_t)q * nst + d] = q;
        }
    for (int d = 0; d <= nst; d++) eoff[(size_t)d] = d * r;
    for (int d = 0; d < nst; d++)
        for (int q = 0; q < r; q++) erec[(size_t)d * r + q] = q * nst + d;
    hipStream_t st = g0.stream;
    T.cl_off.upload(off, st);
    T.cl_step.upload(steps, st);
    T.cl_owner.upload(owner, st);
    T.co_off.upload(eoff, st);
    T.co_rec.upload(erec, st);
    T.co_elim.upload(elim, st);
    T.xbuf.alloc(T.x_doubles());
    T.Dinv.alloc((size_t)7 * B * B);
    T.topDinv.alloc((size_t)B * B);
    T.recR.alloc((size_t)r * nst * B);
    T.recW.alloc((size_t)r * nst * B);
    T.Ttop.alloc((size_t)r * 3);
    T.lam.alloc((size_t)T.npad * 3);
    T.dead.alloc(1);
    IRH_CHECK(hipMemsetAsync(T.topDinv.p, 0, sizeof(double) * (size_t)B * B, st));
    IRH_CHECK(hipMemsetAsync(T.xbuf.p, 0, sizeof(double) * T.x_doubles(), st));
    IRH_CHECK(hipStreamSynchronize(st));  // the host vectors go away
}
#define IRH_BCR_DISPATCH(fn, ...)              \
    switch (g.bcr_B) {                         \
    case 8: fn<8>(__VA_ARGS__); break;         \
    case 16: fn<16>(__VA_ARGS__); break;       \
    case 24: fn<24>(__VA_ARGS__); break;       \
    default: fn<32>(__VA_ARGS__); break;       \
    }
void bcr_shard_reduce(Graph &g, BcrTop &T, int rank) {
    bcr_alloc(g);
    IRH_BCR_DISPATCH(bcr_shard_reduce_t, g, T, rank)
}
void bcr_top_solve(Graph &g, BcrTop &T) { IRH_BCR_DISPATCH(bcr_top_solve_t, g, T) }
void bcr_shard_back(Graph &g, BcrTop &T, int rank) {
    IRH_BCR_DISPATCH(bcr_shard_back_t, g, T, rank)
    g.stats.direct_solves += 1;
}
void bcr_shard_closures_forward(Graph &g, BcrTop &T, int rank) { IRH_BCR_DISPATCH(bcr_shard_closures_forward_t, g, T, rank) }
void bcr_top_solve_closures(Graph &g, BcrTop &T) { IRH_BCR_DISPATCH(bcr_top_solve_closures_t, g, T) }
void bcr_shard_closures_correct(Graph &g, BcrTop &T) { IRH_BCR_DISPATCH(bcr_shard_closures_correct_t, g, T) }

// levels[0] values, diagonal and right-hand side (assemble_values) -> g.X. Asynchronous; only >= 0 launches one
// kernel alone (bench: 0.. the reductions, 100.. the ways back).
int bcr_solve(Graph &g, int only) {
    if (!g.bcr_B) return IROTAVG_ERR_BAD_ARG;
    bcr_alloc(g);
    switch (g.bcr_B) {
    case 8: bcr_run_all<8>(g, only); break;
    case 12: bcr_run_all<12>(g, only); break;
    case 16: bcr_run_all<16>(g, only); break;
    case 20: bcr_run_all<20>(g, only); break;
    case 24: bcr_run_all<24>(g, only); break;
    case 28: bcr_run_all<28>(g, only); break;
    case 32: bcr_run_all<32>(g, only); break;
    default: return IROTAVG_ERR_BAD_ARG;
    }
    if (only < 0) g.stats.direct_solves += 1;
    return IROTAVG_OK;
}

// ||b - A x||^2 and ||b||^2 per coordinate of the level-0 system as it stands (values and diagonal of the last assembly,
// its right-hand side, g.X): a row per thread walks its SELL-64 row like bcr_gather_row does. The loop closures' entries
// are part of these rows, so the Woodbury correction is checked as well.
__global__ __launch_bounds__(256) void k_bcr_residual(int n, const int *__restrict__ sl_off, const int *__restrict__ col,
                                                       const double *__restrict__ val, const double *__restrict__ diag,
                                                       const double4 *__restrict__ rhs, const double4 *__restrict__ X,
                                                       double *__restrict__ part) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    double r0 = 0.0, r1 = 0.0, r2 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0;
    if (row < n) {
        const int sl = row >> 6, ln = row & 63;
        const int o0 = sl_off[sl], w = sl_off[sl + 1] - o0;
        const v2i *__restrict__ cp = reinterpret_cast<const v2i *>(col) + (size_t)(o0 / 2) * 64 + ln;
        const v2d *__restrict__ vp = reinterpret_cast<const v2d *>(val) + (size_t)(o0 / 2) * 64 + ln;
        const double4 x = X[row], bb = rhs[row];
        const double d = diag[row];
        double s0 = d * x.x, s1 = d * x.y, s2 = d * x.z;
        for (int q = 0; q < w / 2; q++) {
            const v2i cc = cp[(size_t)q * 64];
            const v2d vv = vp[(size_t)q * 64];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const double v = h ? vv.y : vv.x;
                const int c = h ? cc.y : cc.x;
                if (v != 0.0) {  // (zero: padding)
                    const double4 xc = X[c];
                    s0 += v * xc.x;
                    s1 += v * xc.y;
                    s2 += v * xc.z;
                }
            }
        }
        r0 = (bb.x - s0) * (bb.x - s0);
        r1 = (bb.y - s1) * (bb.y - s1);
        r2 = (bb.z - s2) * (bb.z - s2);
        b0 = bb.x * bb.x;
        b1 = bb.y * bb.y;
        b2 = bb.z * bb.z;
    }
    block_sum3_store(r0, r1, r2, part + 8 * blockIdx.x);
    __syncthreads();
    block_sum3_store(b0, b1, b2, part + 8 * blockIdx.x + 4);
}

// a direct solve has no residual test of its own. On demand (one pass over level 0, a host round trip):
// ||b - A x|| / ||b|| per coordinate of the handle's most recent direct solve -> relres[3] and stats.last_relres.
int bcr_residual(Graph &g, double *relres) {
    if (!g.bcr_B || g.stats.direct_solves == 0 || g.levels.empty()) return IROTAVG_ERR_BAD_ARG;
    // not for a shard (its rows sit behind the ghost views, and a rank's rows are no system of their own) ...
    if (g.ng > 0 || g.bcr_shard) return IROTAVG_ERR_BAD_ARG;
    // ... and after a GUARDED solve (run_irls: a dead pivot of the band part -> conjugate gradients on the full operator,
    // preconditioned by the regularised direct solve) the right-hand side array holds that iteration's residual vector,
    // not b: what it reached is in stats.last_relres already (its own convergence test); nothing to evaluate here
    if (g.bcr_last_guarded) {
        for (int c = 0; c < 3; c++) relres[c] = g.stats.last_relres[c];
        return IROTAVG_OK;
    }
    Level &L0 = g.levels[0];
    const int grid = (L0.n + 255) / 256;
    DevBuf<double> part;
    part.alloc((size_t)grid * 8);
    hipLaunchKernelGGL(k_bcr_residual, dim3(grid), dim3(256), 0, g.stream, L0.n, L0.sl_off.p, L0.col.p, L0.val.p, L0.diag.p,
                       L0.b.p, g.X.p, part.p);
    IRH_CHECK(hipGetLastError());
    std::vector<double> h((size_t)grid * 8);
    IRH_CHECK(hipMemcpyAsync(h.data(), part.p, sizeof(double) * h.size(), hipMemcpyDeviceToHost, g.stream));
    IRH_CHECK(hipStreamSynchronize(g.stream));
    double rr[3] = {0, 0, 0}, bb[3] = {0, 0, 0};
    for (int b = 0; b < grid; b++)
        for (int c = 0; c < 3; c++) {
            rr[c] += h[(size_t)b * 8 + c];
            bb[c] += h[(size_t)b * 8 + 4 + c];
        }
    for (int c = 0; c < 3; c++) {
        relres[c] = bb[c] > 0.0 ? std::sqrt(rr[c] / bb[c]) : (rr[c] > 0.0 ? HUGE_VAL : 0.0);
        g.stats.last_relres[c] = relres[c];
    }
    return IROTAVG_OK;
}

int bcr_info(Graph &g, int64_t *out, int cap) {
    int k = 0;
    auto put = [&](int64_t v) {
        if (k < cap) out[k++] = v;
    };
    put(g.bcr_B);
    if (!g.bcr_B) return k;
    bcr_alloc(g);
    put((int64_t)g.bcr->lev.size());
    for (const BcrLevel &L : g.bcr->lev) {
        put(L.nb);
        put(L.nch);
        put(L.nred);
    }
    put(g.bcr->nfar);
    return k;
}

// development aid: a whole solve with stamps inside the single-launch upper reduction (IROTAVG_BCR_DBG & 128): out[32 i + k] =
// shader clocks of workgroup 0 at level i of the launch relative to the launch's first stamp -- k < 16: the body's phases,
// 16: level entered, 17: the wait for the level below is over, 18: body done, 19: counted; -1: not stamped
int bcr_stamps_up(Graph &g, double *out) {
    if (!g.bcr_B) return IROTAVG_ERR_BAD_ARG;
    bcr_alloc(g);
    BcrState &S = *g.bcr;
    const char *wgs = getenv("IROTAVG_BCR_STAMP_CHUNK");
    struct DbgScope {
        BcrState &S;
        ~DbgScope() { S.dbg_word = -1; }
    } dbg_scope{S};
    S.dbg_word = 128 + 256 * (wgs ? atoi(wgs) : 0);
    const size_t cnt = (size_t)std::max(S.lev[0].nch * 16, 32 * kUpLevels);
    if (!S.stamps.p) S.stamps.alloc(cnt);
    IRH_CHECK(hipMemsetAsync(S.stamps.p, 0, sizeof(long long) * cnt, g.stream));
    const int rc = bcr_solve(g, -1);
    S.dbg_word = -1;
    if (rc != IROTAVG_OK) return rc;
    long long h[32 * kUpLevels];
    IRH_CHECK(hipMemcpyAsync(h, S.stamps.p, sizeof(h), hipMemcpyDeviceToHost, g.stream));
    IRH_CHECK(hipStreamSynchronize(g.stream));
    bcr_up_release(g);
    for (int k = 0; k < 32 * kUpLevels; k++) out[k] = !h[k] ? -1.0 : (k & 31) >= 20 ? (double)h[k] : (double)(h[k] - h[16]);
    return S.up_used ? IROTAVG_OK : IROTAVG_ERR_BAD_ARG;
}

// development aid: the reduction of `level` once with stamps on; out[0..16) = shader clocks of chunk `chunk` relative to
// its first stamp
int bcr_stamps(Graph &g, int level, int chunk, double *out) {
    if (!g.bcr_B) return IROTAVG_ERR_BAD_ARG;
    bcr_alloc(g);
    BcrState &S = *g.bcr;
    if (level < 0 || level >= (int)S.lev.size() || chunk < 0 || chunk >= S.lev[level].nch) return IROTAVG_ERR_BAD_ARG;
    struct DbgScope {
        BcrState &S;
        ~DbgScope() { S.dbg_word = -1; }
    } dbg_scope{S};
    S.dbg_word = 64;
    if (!S.stamps.p) S.stamps.alloc((size_t)S.lev[0].nch * 16);
    IRH_CHECK(hipMemsetAsync(S.stamps.p, 0, sizeof(long long) * (size_t)S.lev[0].nch * 16, g.stream));
    const int rc = bcr_solve(g, level);
    S.dbg_word = -1;
    if (rc != IROTAVG_OK) return rc;
    long long h[16];
    IRH_CHECK(hipMemcpyAsync(h, S.stamps.p + (size_t)chunk * 16, sizeof(h), hipMemcpyDeviceToHost, g.stream));
    IRH_CHECK(hipStreamSynchronize(g.stream));
    for (int k = 0; k < 16; k++) out[k] = h[k] ? (double)(h[k] - h[0]) : -1.0;
    return IROTAVG_OK;
}

// ||b - Ax||^2 and ||b||^2 per coordinate from A x (partial sums in k_bcr_residual's layout)
__global__ __launch_bounds__(256) void k_bcr_resid_norms(int n, const double4 *__restrict__ b, const double4 *__restrict__ Ax,
                                                          double *__restrict__ part) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    double r0 = 0.0, r1 = 0.0, r2 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0;
    if (row < n) {
        const double4 bb = b[row], a = Ax[row];
        r0 = (bb.x - a.x) * (bb.x - a.x);
        r1 = (bb.y - a.y) * (bb.y - a.y);
        r2 = (bb.z - a.z) * (bb.z - a.z);
        b0 = bb.x * bb.x;
        b1 = bb.y * bb.y;
        b2 = bb.z * bb.z;
    }
    block_sum3_store(r0, r1, r2, part + 8 * blockIdx.x);
    __syncthreads();
    block_sum3_store(b0, b1, b2, part + 8 * blockIdx.x + 4);
}

// The verdict on a direct solve with closures, taken on the device: no dead pivot of the band part AND a residual of the
// FULL system (k_bcr_residual's partial sums: one pass over level 0) within tol. The Woodbury form is exact in exact
// arithmetic only: where robust weights leave a stretch of the band nearly free and the closures hold it, Y = A_b^-1 b is
// huge along that stretch and the correction cancels it (fuzz seed 21 case 46: Welsch, 365 closures, weights^2 down to
// 1e-8 -- relative residual 4e-7 where the band-only solve reaches 1e-12). flags[1] = 1: the residual was the reason.
__global__ __launch_bounds__(256) void k_bcr_gate(int *__restrict__ dead, int *__restrict__ flags,
                                                  const double *__restrict__ part, int nparts, double tol2,
                                                  int *__restrict__ skip_out, int *__restrict__ flags_copy) {
    __shared__ double sh[4][6];
    const int t = threadIdx.x;
    double a[6] = {0, 0, 0, 0, 0, 0};
    if (part)
        for (int b = t; b < nparts; b += 256) {
            for (int c = 0; c < 3; c++) {
                a[c] += part[(size_t)b * 8 + c];
                a[3 + c] += part[(size_t)b * 8 + 4 + c];
            }
        }
    // (the sums of the 256 threads: by wave shuffles, then over the four waves -- one thread adding 1536 LDS words took 8 us)
    for (int c = 0; c < 6; c++) {
        double v = a[c];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
        if ((t & 63) == 0) sh[t >> 6][c] = v;
    }
    __syncthreads();
    if (t != 0) return;
    bool ok = true;
    if (part) {
        for (int c = 0; c < 6; c++) a[c] = sh[0][c] + sh[1][c] + sh[2][c] + sh[3][c];
        for (int c = 0; c < 3; c++) ok = ok && (a[c] <= tol2 * a[3 + c]);  // (NaN: not ok)
    }
    const int d = dead ? dead[0] : 0;
    flags[FL_DONE] = d == 0 && ok ? 1 : 0;
    flags[FL_ITERS] = d == 0 && !ok ? 1 : 0;
    flags[3] = d;
    // (round 6) the same verdict as a skip word for k_weights_then_residual, and the four flag words once more behind the
    // score's partial sums, so that ONE publication -- that kernel's first workgroup -- carries score and verdict
    if (dead) dead[0] = 0;  // (read: the next reduction counts from zero without a memset in front of it, BcrState::dead_clean)
    if (skip_out) skip_out[0] = d == 0 && ok ? 0 : 1;
    if (flags_copy) {
        flags_copy[FL_DONE] = d == 0 && ok ? 1 : 0;
        flags_copy[FL_ITERS] = d == 0 && !ok ? 1 : 0;
        flags_copy[FL_STALE] = 0;
        flags_copy[3] = d;
    }
}
const int *bcr_gate_skip_word(Graph &g) {
    BcrState *S = g.bcr.get();
    if (!S) return nullptr;
    if (!S->gate_skip.p) {
        S->gate_skip.alloc(1);
        S->gate_skip.zero(g.stream);
    }
    return S->gate_skip.p;
}
void bcr_gate(Graph &g, double *flags_copy) {
    BcrState *S = g.bcr.get();
    int *dead = S && S->nfar > 0 ? S->dead.p : nullptr;
    if (dead) S->dead_clean = true;  // (the gate kernel leaves the counter at zero)
    const double *part = nullptr;
    int grid = 0;
    // (IROTAVG_BCR_NO_RESIDUAL_GATE: the dead-pivot gate alone, as until round 5)
    static const bool no_res = getenv("IROTAVG_BCR_NO_RESIDUAL_GATE") != nullptr;
    if (S && S->nfar > 0 && g.ng == 0 && !g.bcr_shard && !no_res) {
        // A x by the iterative solver's SpMV (13 us at 2M edges; a row per thread walking its SELL row, k_bcr_residual,
        // took 45), then b - A x and the two norms in one small pass over the rows
        Level &L0 = g.levels[0];
        grid = (L0.n + 255) / 256;
        if (S->res_part.n < (size_t)grid * 8) {
            S->res_part.alloc((size_t)grid * 8);
            S->res_zero.alloc(FL_COUNT);
            S->res_zero.zero(g.stream);  // (the SpMV skips its work behind a FL_DONE that is set: it gets words that never are)
        }
        launch_spmv(g, g.X.p, nullptr, S->res_zero.p);
        hipLaunchKernelGGL(k_bcr_resid_norms, dim3(grid), dim3(256), 0, g.stream, L0.n, L0.b.p, g.AP.p, S->res_part.p);
        part = S->res_part.p;
    }
    // (IROTAVG_BCR_FAKE_GIVE_UP, tests: no residual passes the gate)
    const double gate = std::max(g.opt.pcg_rtol, kBcrGateTolMin);
    const double tol = getenv("IROTAVG_BCR_FAKE_GIVE_UP") ? -1.0 : gate * gate;
    int *skip_out = (S && flags_copy) ? const_cast<int *>(bcr_gate_skip_word(g)) : nullptr;
    hipLaunchKernelGGL(k_bcr_gate, dim3(1), dim3(256), 0, g.stream, dead, g.flags.p, part, grid, tol, skip_out,
                       reinterpret_cast<int *>(flags_copy));
}

int bcr_closures(Graph &g) { return g.bcr_B ? (int)g.bcr_far_e.size() : 0; }

// partial sums of the score a solve with Graph::bcr_apply leaves in part_score (one double per workgroup of the two ways
// back), or 0 when such a solve cannot make the step itself: closures, a shard, one level, more slots than the pinned block
// has doubles for
int bcr_apply_slots(Graph &g) {
    if (!g.bcr_B || !g.bcr_far_e.empty() || g.bcr_shard || g.ng != 0 || getenv("IROTAVG_BCR_NO_FUSED_BACK") ||
        getenv("IROTAVG_BCR_NO_APPLY"))
        return 0;
    bcr_alloc(g);
    const BcrState &S = *g.bcr;
    if (S.lev.size() < 2 || S.nfar != 0) return 0;
    // (level 0's chunks and, when there is a level of chunks above them, level 1's: a mixed level 1 writes solution rows)
    const int slots = S.lev[0].nch + (S.top16 && S.lev.size() == 2 ? 0 : S.lev[1].nch);
    return slots <= 4 * kMaxParts ? slots : 0;
}

const int *bcr_fail_word(Graph &g) {
    if (!g.bcr || !g.bcr->up_used || g.bcr->up_cnt.n < (size_t)kUpLevels + 1) return nullptr;
    return reinterpret_cast<const int *>(g.bcr->up_cnt.p + kUpLevels);
}
bool bcr_up_failed(Graph &g) {
    const int *w = bcr_fail_word(g);
    if (!w) return false;
    int h = 0;
    if (hipMemcpyAsync(&h, w, sizeof(int), hipMemcpyDeviceToHost, g.stream) != hipSuccess) return false;
    if (hipStreamSynchronize(g.stream) != hipSuccess) return false;
    if (!h) return false;
    (void)hipMemsetAsync(const_cast<int *>(w), 0, sizeof(int), g.stream);
    g.bcr_no_fused_up = true;
    g.stats.direct_up_fallbacks += 1;
    bcr_up_release(g);
    return true;
}

int bcr_levels(Graph &g) {
    if (!g.bcr_B) return 0;
    bcr_alloc(g);
    return (int)g.bcr->lev.size();
}

// Called BEFORE the build (capi.cpp): a handle that solves directly gets no multigrid hierarchy at all.
// Decides whether the handle's solves run here: one GPU, every edge between two free views within 32 views
// (ral's I is 0-based; row = view - f) except for at most kBcrMaxFar long-range edges (loop closures: Woodbury
// correction, bcr_solve), enough rows for the hierarchy of the iterative solver to exist at all (smaller graphs
// are one dense level = a direct solve already). opt.band_direct: 0 choose, 1 whenever the band allows, -1 never;
// IROTAVG_BAND_DIRECT overrides the option.
constexpr int kBcrMaxFar = 2048;

// Is the BAND part of the operator positive definite -- is every free view tied to a fixed one through edges that are
// not closures (free-free edges within 32 views or between neighbouring blocks, and the edges from a fixed view that
// make_A keeps)? Exact (union-find over the free views); the callers try the cheap sufficient rule first.
bool bcr_band_part_anchored(int64_t m, int f, int64_t nu, int B, const int32_t *I) {
    std::vector<int> parent((size_t)nu);
    for (int64_t r = 0; r < nu; r++) parent[(size_t)r] = (int)r;
    std::vector<uint8_t> anch((size_t)nu, 0);
    auto find = [&](int x) {
        while (parent[(size_t)x] != x) {
            parent[(size_t)x] = parent[(size_t)parent[(size_t)x]];
            x = parent[(size_t)x];
        }
        return x;
    };
    for (int64_t k = 0; k < m; k++) {
        const int i = I[2 * k], j = I[2 * k + 1];
        if (i >= f && j >= f) {
            if (i == j) continue;
            if (std::abs(i - j) > 32 && std::abs((i - f) / B - (j - f) / B) >= 2) continue;  // a closure
            const int a = find(i - f), b = find(j - f);
            if (a != b) parent[(size_t)std::max(a, b)] = std::min(a, b);
        } else if (i < f && j >= f) {
            anch[(size_t)(j - f)] = 1;
        }
    }
    for (int64_t r = 0; r < nu; r++)
        if (anch[(size_t)r]) anch[(size_t)find((int)r)] = 1;
    for (int64_t r = 0; r < nu; r++)
        if (!anch[(size_t)find((int)r)]) return false;
    return true;
}

void bcr_plan(Graph &g, const int32_t *I) {
    g.bcr_B = 0;
    g.band0 = -1;
    g.bcr_far_i.clear();
    g.bcr_far_j.clear();
    g.bcr_far_e.clear();
    int mode = g.opt.band_direct;
    if (const char *e = std::getenv("IROTAVG_BAND_DIRECT")) mode = std::atoi(e);
    if (mode < 0 || g.ng > 0 || g.no < 1) return;
    const int f = g.f;
    std::atomic<int> band(0), bandall(0);
    std::atomic<long long> nfar(0);
    std::atomic<bool> bad(false);
    const int64_t nt = g.n_total;
    parallel_for(g.m, 65536, [&](int64_t k0, int64_t k1, int) {
        int bmax = 0, ball = 0;
        long long far = 0;
        for (int64_t k = k0; k < k1; k++) {
            const int i = I[2 * k], j = I[2 * k + 1];
            if (i < 0 || j < 0 || i >= nt || j >= nt) {
                bad = true;
                return;
            }
            if (i >= f && j >= f) {
                const int d = std::abs(i - j);
                ball = std::max(ball, d);
                if (d <= 32) bmax = std::max(bmax, d);
                else far++;
            }
        }
        int cur = band.load();
        while (bmax > cur && !band.compare_exchange_weak(cur, bmax)) {
        }
        cur = bandall.load();
        while (ball > cur && !bandall.compare_exchange_weak(cur, ball)) {
        }
        nfar += far;
    });
    if (bad.load()) return;
    g.band0 = bandall.load();
    // closures cost the direct path 0.06 ms + 0.55 us r + 0.15 ns r^2 per solve (forward eliminations, inversion of the
    // r x r Woodbury system, its assembly; measured at r = 12 ... 1000), the iterative solver ~0.9 ms per solve at 3000
    // views with loop edges and ~2.2 ms at 20k ... 100k: up to 2048 closures stay direct, up to 1024 on small graphs
    if (nfar.load() > (g.no < 8192 ? kBcrMaxFar / 2 : kBcrMaxFar)) return;
    if (mode == 0 && g.no <= 2048) return;
    const int b0 = band.load();
    // blocks of 8, 12, ... 32 rows: the smallest multiple of four that holds the band (a shard of a sharded sequence
    // keeps 8 / 16 / 24 / 32: its range is a multiple of 192 rows, dist.hip); IROTAVG_BCR_BLOCK asks for a larger one
    int B = b0 <= 8 ? 8 : (b0 + 3) / 4 * 4;
    if (const char *e = std::getenv("IROTAVG_BCR_BLOCK")) {
        const int want = std::atoi(e);
        if (want >= B && want <= 32 && want % 4 == 0) B = want;
    }
    if (nfar.load() > 0) {
        // long-range edges = those whose endpoints lie in blocks that are not neighbours (an edge of more than 32
        // views between neighbouring blocks is part of the block tridiagonal operator as it is)
        for (int64_t k = 0; k < g.m; k++) {
            const int i = I[2 * k], j = I[2 * k + 1];
            if (i >= f && j >= f && std::abs(i - j) > 32 && std::abs((i - f) / B - (j - f) / B) >= 2) {
                g.bcr_far_i.push_back(i - f);
                g.bcr_far_j.push_back(j - f);
                g.bcr_far_e.push_back((int)k);
            }
        }
        bool refuse = !g.bcr_far_e.empty() && getenv("IROTAVG_BCR_NO_CLOSURES");
        if (!g.bcr_far_e.empty() && !refuse) {
            // The Woodbury correction needs the BAND part alone to be positive definite: every free view must be
            // tied to a fixed one through band edges. Sufficient and cheap: every view has a band edge to an earlier
            // view or an edge to a fixed view that the IRLS system keeps (make_A drops an edge whose SECOND endpoint
            // is fixed, ral/l1_irls.cpp:770-771). A sequence whose band part falls apart (a stretch that only a
            // closure ties to the rest) is left to the iterative solver.
            std::vector<uint8_t> ok((size_t)g.no, 0);
            parallel_for(g.m, 65536, [&](int64_t k0, int64_t k1, int) {
                for (int64_t k = k0; k < k1; k++) {
                    const int i = I[2 * k], j = I[2 * k + 1];
                    if (i >= f && j >= f) {
                        if (i != j && std::abs(i - j) <= 32) ok[(size_t)(std::max(i, j) - f)] = 1;  // benign race: all write 1
                    } else if (i < f && j >= f) {
                        ok[(size_t)(j - f)] = 1;
                    }
                }
            });
            for (int r = 0; r < g.no && !refuse; r++) refuse = !ok[(size_t)r];
            // (the rule is sufficient, not necessary -- e.g. the first free view of a sequence whose fixed views lie
            // further on is tied through LATER views only: the exact test decides then)
            if (refuse) refuse = !bcr_band_part_anchored(g.m, f, g.no, B, I);
        }
        if (refuse) {
            g.bcr_far_i.clear();
            g.bcr_far_j.clear();
            g.bcr_far_e.clear();
            return;
        }
    }
    g.bcr_B = B;
}

// ---- the same plan from an edge list that lives on the device (a resident view-graph, resident.hip) --------------
namespace {
// out: [0] widest span <= 32, [1] widest span, [2] index out of range; nfar: spans > 32 between free views
__global__ __launch_bounds__(256) void k_plan_stats(long long m, int f, int nt, const int2 *__restrict__ I,
                                                    const int *__restrict__ relabel, int *__restrict__ out,
                                                    unsigned long long *__restrict__ nfar) {
    __shared__ int sb[3][4];
    __shared__ unsigned long long sf[4];
    int band = 0, ball = 0, bad = 0;
    unsigned long long far = 0;
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < m; k += (long long)gridDim.x * 256) {
        const int2 e = I[k];
        int i = e.x, j = e.y;
        if (i < 0 || j < 0 || i >= nt || j >= nt) {
            bad = 1;
            continue;
        }
        if (relabel) {
            i = relabel[i];
            j = relabel[j];
        }
        if (i >= f && j >= f) {
            const int d = abs(i - j);
            ball = max(ball, d);
            if (d <= 32) band = max(band, d);
            else far++;
        }
    }
    for (int o = 32; o >= 1; o >>= 1) {
        band = max(band, __shfl_xor(band, o));
        ball = max(ball, __shfl_xor(ball, o));
        bad |= __shfl_xor(bad, o);
        far += __shfl_xor(far, o);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
        sb[0][w] = band;
        sb[1][w] = ball;
        sb[2][w] = bad;
        sf[w] = far;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        atomicMax(out, max(max(sb[0][0], sb[0][1]), max(sb[0][2], sb[0][3])));
        atomicMax(out + 1, max(max(sb[1][0], sb[1][1]), max(sb[1][2], sb[1][3])));
        if (sb[2][0] | sb[2][1] | sb[2][2] | sb[2][3]) atomicOr(out + 2, 1);
        const unsigned long long t = sf[0] + sf[1] + sf[2] + sf[3];
        if (t) atomicAdd(nfar, t);
    }
}
// the long-range edges (rows, edge id; any order: the host sorts them by edge id) and the coverage of the band part:
// ok[r] = 1 when row r has a band edge to an earlier view or an edge to a fixed view that the IRLS system keeps
__global__ __launch_bounds__(256) void k_plan_far(long long m, int f, int B, const int2 *__restrict__ I,
                                                  const int *__restrict__ relabel, int cap, int *__restrict__ cnt,
                                                  int *__restrict__ fi, int *__restrict__ fj, int *__restrict__ fe,
                                                  uint8_t *__restrict__ ok) {
    const long long k = (long long)blockIdx.x * 256 + threadIdx.x;
    if (k >= m) return;
    const int2 e = I[k];
    const int i = relabel ? relabel[e.x] : e.x, j = relabel ? relabel[e.y] : e.y;
    if (i >= f && j >= f) {
        const int d = abs(i - j);
        if (d > 32 && abs((i - f) / B - (j - f) / B) >= 2) {
            const int q = atomicAdd(cnt, 1);
            if (q < cap) {
                fi[q] = i - f;
                fj[q] = j - f;
                fe[q] = (int)k;
            }
        }
        if (i != j && d <= 32) ok[max(i, j) - f] = 1;
    } else if (i < f && j >= f) {
        ok[j - f] = 1;
    }
}
__global__ __launch_bounds__(256) void k_plan_uncovered(int no, const uint8_t *__restrict__ ok, int *__restrict__ cnt) {
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r < no && !ok[r]) atomicAdd(cnt, 1);
}
}  // namespace

void bcr_plan_dev(Graph &g, const DevEdgeSrc &src) {
    g.bcr_B = 0;
    g.band0 = -1;
    g.bcr_far_i.clear();
    g.bcr_far_j.clear();
    g.bcr_far_e.clear();
    int mode = g.opt.band_direct;
    if (const char *e = std::getenv("IROTAVG_BAND_DIRECT")) mode = std::atoi(e);
    if (mode < 0 || g.ng > 0 || g.no < 1) return;
    const int f = g.f;
    hipStream_t s = g.stream;
    int *h = reinterpret_cast<int *>(PinPool::get().take());
    struct PinGuard {
        int *p;
        ~PinGuard() { PinPool::get().give(p); }
    } pin_guard{h};
    DevBuf<int> out;  // [0..2] the statistics, [4..5] the far count (64 bit), [6] far-list length, [7] uncovered rows
    out.alloc(8);
    out.zero(s);
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>(1024, (g.m + 255) / 256));
    hipLaunchKernelGGL(k_plan_stats, dim3(grid), dim3(256), 0, s, (long long)g.m, f, (int)g.n_total, src.I, src.relabel, out.p,
                       reinterpret_cast<unsigned long long *>(out.p + 4));
    IRH_CHECK(hipMemcpyAsync(h, out.p, 8 * sizeof(int), hipMemcpyDeviceToHost, s));
    IRH_CHECK(hipStreamSynchronize(s));
    if (h[2]) return;
    g.band0 = h[1];
    const long long nfar = (long long)*reinterpret_cast<unsigned long long *>(h + 4);
    if (nfar > (g.no < 8192 ? kBcrMaxFar / 2 : kBcrMaxFar)) return;
    if (mode == 0 && g.no <= 2048) return;
    const int b0 = h[0];
    int B = b0 <= 8 ? 8 : (b0 + 3) / 4 * 4;
    if (const char *e = std::getenv("IROTAVG_BCR_BLOCK")) {
        const int want = std::atoi(e);
        if (want >= B && want <= 32 && want % 4 == 0) B = want;
    }
    if (nfar > 0) {
        DevBuf<int> fl;
        DevBuf<uint8_t> ok;
        fl.alloc((size_t)3 * kBcrMaxFar);
        ok.alloc((size_t)g.no);
        ok.zero(s);
        hipLaunchKernelGGL(k_plan_far, dim3((unsigned)((g.m + 255) / 256)), dim3(256), 0, s, (long long)g.m, f, B, src.I,
                           src.relabel, kBcrMaxFar, out.p + 6, fl.p, fl.p + kBcrMaxFar, fl.p + 2 * kBcrMaxFar, ok.p);
        hipLaunchKernelGGL(k_plan_uncovered, dim3((unsigned)((g.no + 255) / 256)), dim3(256), 0, s, g.no, ok.p, out.p + 7);
        IRH_CHECK(hipMemcpyAsync(h, out.p + 6, 2 * sizeof(int), hipMemcpyDeviceToHost, s));
        IRH_CHECK(hipStreamSynchronize(s));
        const int nl = h[0], uncovered = h[1];
        if (nl > kBcrMaxFar) return;  // (cannot happen: nl <= nfar)
        if (nl > 0) {
            if (uncovered > 0 || getenv("IROTAVG_BCR_NO_CLOSURES")) return;  // see bcr_plan: the band part must be SPD alone
            std::vector<int> hl((size_t)3 * kBcrMaxFar);
            IRH_CHECK(hipMemcpyAsync(hl.data(), fl.p, sizeof(int) * hl.size(), hipMemcpyDeviceToHost, s));
            IRH_CHECK(hipStreamSynchronize(s));
            std::vector<int> order((size_t)nl);
            for (int q = 0; q < nl; q++) order[(size_t)q] = q;
            std::sort(order.begin(), order.end(),
                      [&](int a, int b) { return hl[(size_t)2 * kBcrMaxFar + a] < hl[(size_t)2 * kBcrMaxFar + b]; });
            for (int q : order) {  // edge order, as the host plan lists them
                g.bcr_far_i.push_back(hl[(size_t)q]);
                g.bcr_far_j.push_back(hl[(size_t)kBcrMaxFar + q]);
                g.bcr_far_e.push_back(hl[(size_t)2 * kBcrMaxFar + q]);
            }
        }
    }
    g.bcr_B = B;
}

}  // namespace irh

// qfa_grads_t.h -- pass 2 (gradients) for N_h = 9..16 in its PIXEL-RESIDENT form: a wave owns 16 pixels and walks the spectra.
//
// k_grads_x (qfa_grads_x.h) keeps the per-spectrum operands of 64 spectra in registers and walks the pixel axis: every
// tile step ends in a contraction of W with F on the VALU, 34 16-byte LDS stores of partial sums per wave, a flush that
// reads them back and 0.54 GB of float atomics per launch at c3 -- and the kernel is bound by the instructions it
// issues ("What binds pass 2").  Every output of pass 2 is a sum over SPECTRA per pixel, so here
// the roles of the two axes are exchanged:
//   * a wave owns ONE 16-pixel tile for the whole launch: the stage-1 image of its pixels (B operand: two float16 pieces,
//     40 registers -- three bf16 pieces, 60, until round 5), their Psi / omega / factored-z terms, and the running sums --
//     W[px][a][b] = sum_s Z_s[a][b] beta[s][px] as 16 MFMA accumulator tiles (64 registers), the gamma term, gPsi, gOmega, sumA,
//     the count;
//   * the per-spectrum operands of a group of 16 spectra (the "state": [Cinv' | y] pieces for stage 1, Z and p pieces
//     for stage 3, 30 KiB (52 with bf16 pieces), written once per launch by k_solve) stream through an LDS ring by LDS-DMA,
//     shared by the workgroup's 8 waves (8 tiles = 128 pixels);
//   * beta / gamma never leave the lane: stage 2 leaves them in exactly the layout stage 3's B operand wants;
//   * F enters once, at the end: accF[px][b] = sum_a F[px][a] W[px][a][b] + the gamma term, then ONE atomic (or slab
//     store) per output element and spectra range -- no partial sums in LDS, no flush in the loop.
// Per (16 spectra x 16 pixels) a wave issues 18 + 35 MFMAs (round 5: two float16 pieces and three products per contraction,
// qfa_common.h "float16 pieces", qfa_gt_layout.h QFA_GT_F16S1 / QFA_GT_F16S3; 36 + 51 with bf16 pieces), 12 + 19 ds_read_b128 for
// their streamed operands, stage 2 of four elements per lane, one float16 split (beta) and one bf16 split (gamma).
//
// Workgroup = 512 threads = 8 waves = 8 tiles (strided over the pixel axis: tile pb + PB w, so that every workgroup
// gets its share of blue tiles) x one RANGE of spectra groups; PB pixel blocks x R ranges, mapped to the grid so that the
// workgroups of one range sit on one XCD (GtPlan) and walk the same state in step: it is read from HBM once and from that
// XCD's L2 by the others (measured: 3.9 GB fetched per launch at c3 against 3.6 GB of spectra).
// The schedule of a step (one barrier per group, the two waves of a SIMD a stage apart) is described at the walk below.
// The spectra (delta, sigma, zabs rows -- or the per-spectrum factors of the factored-z form: 64 bytes per row and tile;
// masks 16) are staged per wave, two groups ahead, as in k_grads_x.  All DMA is asm (untracked); every wave waits for all of its
// requests in front of the step's barrier (round 5).  Ragged tiles (the last tile of a pixel axis that is no multiple of 16, the
// tile that straddles the end of the blue side in the zabs form) stage 4-byte pieces through the general path.
#pragma once
#ifndef QFA_GT_SETPRIO
#define QFA_GT_SETPRIO 2   // s_setprio around the MFMA stages of k_grads_t: the wave of a SIMD that is in stage 3 (or 1) issues ahead of its
                           // partner's VALU stage -- pass 2 at c3 2.24 -> 2.16 ms, DESI shape 1.23 -> 1.19 (levels 1..3 alike; stage 3 carries it)
#endif
#ifndef QFA_GT_PRIO_STAGES
#define QFA_GT_PRIO_STAGES 3   // bit 0: stage 1, bit 1: stage 3
#endif
#include "qfa_common.h"
#include "qfa_xdl_kernels.h"
#include "qfa_gt_layout.h"        // GTT, build_state

// the split of the launch: tiles, pixel blocks, ranges of spectra groups (host and device agree through this struct)
struct GtPlan {
    int T16, PB, R, gpr;          // wave tiles (GTT::PXW pixels each); pixel blocks of 8 wave tiles; ranges; groups of 16 spectra per range
    int nxcd;                     // XCDs the hardware deals workgroups to, round-robin (hipDeviceAttributeNumberOfXccs: 8 on MI355X
                                  // in SPX mode, fewer in the partitioned modes; qfa_host.h xcd_count)
    // Work item i = r PB + pb (range-major).  Workgroup b runs item (b % nxcd) ceil(PB R / nxcd) + b / nxcd: workgroups go to the
    // XCDs round-robin (b % nxcd), so every XCD gets a contiguous run of items -- the pixel blocks of ONE or two ranges, which walk
    // the same state in step and share it in that XCD's L2, whatever R is (the grid is padded to a multiple of nxcd)
    __host__ __device__ int per_xcd() const { return (PB * R + nxcd - 1) / nxcd; }
    __host__ __device__ int items() const { return nxcd * per_xcd(); }
};

// ------------------------------------------------------------------------------------------------
// k_prep_pgt : F, Psi, omega (+ the per-pixel factors of the factored-z form) -> the image of every 16-pixel tile
// ------------------------------------------------------------------------------------------------
template <int KP>
__device__ __forceinline__ void prep_pgt_body(int bid, const float *__restrict__ F, const float *__restrict__ Psi,
                                              const float *__restrict__ omega, const ZPSrc &ZP, int Npix, int Nb, int Nh,
                                              unsigned char *__restrict__ PGT) {
    using GT = GTT<KP>;
    unsigned char *tile = PGT + (size_t)bid * GT::TILE_B;
    // tile bid = TPW wt + j of wave tile wt: column lo <-> pixel PXW wt + TPW lo + j
    const int p0 = GT::PXW * (bid / GT::TPW) + bid % GT::TPW;
    auto pixel = [&](int lo) { return p0 + GT::TPW * lo; };
    __shared__ float f[16][KP + 1];
    __shared__ float tsc[16][3];                                     // F16S1: the pixel's power of two t, 1 / t^2, 1 / t
    for (int i = threadIdx.x; i < 16 * KP; i += 256) {
        const int px = i / KP, a = i % KP;
        f[px][a] = (pixel(px) < Npix && a < Nh) ? F[(size_t)pixel(px) * Nh + a] : 0.f;
    }
    __syncthreads();
    if (GT::F16S1 && threadIdx.x < 16) {
        // t f_a in [2^6, 2^7) for the pixel's largest |f_a|: the pair products t^2 f_a f_b stay below 2^14 (float16: 65 504)
        float mx = 0.f;
        for (int a = 0; a < KP; ++a) mx = fmaxf(mx, fabsf(f[threadIdx.x][a]));
        int e = 7;
        if (mx > 0.f && mx < 3.0e38f) (void)frexpf(mx, &e);          // mx = m 2^e, m in [0.5, 1)
        e = e < -50 ? -50 : (e > 60 ? 60 : e);
        tsc[threadIdx.x][0] = ldexpf(1.f, 7 - e);
        tsc[threadIdx.x][1] = ldexpf(1.f, 2 * (e - 7));
        tsc[threadIdx.x][2] = ldexpf(1.f, e - 7);
    }
    if (GT::F16S1) __syncthreads();
    for (int i = threadIdx.x; i < GT::NKQ * 64; i += 256) {
        const int lane = i & 63, ks = i >> 6;
        const int px = lane & 15, g = lane >> 4;
        const float t1 = GT::F16S1 ? tsc[px][0] : 1.f, t2 = t1 * t1;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int q = 32 * ks + 8 * g + j;                       // pair index, or KK2-block slot of F
            float x = 0.f;
            if (q < GT::KK2) {
                int a = 0;
                while (a + 1 < KP && pair_index(a + 1, a + 1, KP) <= q) ++a;
                const int b = a + (q - pair_index(a, a, KP));
                x = f[px][a] * f[px][b] * t2;                        // (a power of two: exact)
            } else {
                const int a = q - (32 * (GT::NKQ - 1) + GT::YOFF);
                if (a >= 0 && a < KP) x = f[px][a] * t1;
            }
            v[j] = x;
        }
        unsigned char *dst = tile + ks * GT::BLK_B + lane * 16;
        if constexpr (GT::F16S1) {
            u32x4 ph, pm;
            split8h(v, ph, pm);
            *reinterpret_cast<u32x4 *>(dst) = ph;
            *reinterpret_cast<u32x4 *>(dst + 1024) = pm;
        } else {
            u32x4 ph, pm, pl;
            split8(v, ph, pm, pl);
            *reinterpret_cast<u32x4 *>(dst) = ph;
            *reinterpret_cast<u32x4 *>(dst + 1024) = pm;
            *reinterpret_cast<u32x4 *>(dst + 2048) = pl;
        }
    }
    if (threadIdx.x < 128) {
        const int j = threadIdx.x, px = pixel(j & 15);
        float v = 0.f;
        if (j < 16) v = px < Npix ? Psi[px] : 0.f;
        else if (j < 32) v = px < Nb ? omega[px] : 0.f;
        else if (j < 80 && ZP.on() && px < Nb) {
            const float4 q = ZP.at(px);
            v = j < 48 ? q.x : (j < 64 ? q.y : q.z);
        } else if (GT::F16S1 && j >= GT::PAR_IT2 && j < GT::PAR_SBETA) v = tsc[j & 15][j < GT::PAR_IT1 ? 1 : 2];
        else if (GT::F16S3 && j >= GT::PAR_SBETA) v = f16_weight_scale(px < Npix ? Psi[px] : 1.f);       // beta <= 1 / Psi (qfa_common.h)
        reinterpret_cast<float *>(tile + GT::OFF_PAR)[j] = v;
    }
    float *fr = reinterpret_cast<float *>(tile + GT::OFF_F);
    for (int i = threadIdx.x; i < 16 * KP; i += 256) fr[i] = f[i / KP][i % KP];
}
template <int KP>
__global__ __launch_bounds__(256) void k_prep_pgt(const float *__restrict__ F, const float *__restrict__ Psi,
                                                  const float *__restrict__ omega, const float4 *__restrict__ ZP,
                                                  int Npix, int Nb, int Nh, unsigned char *__restrict__ PGT) {
    prep_pgt_body<KP>(blockIdx.x, F, Psi, omega, zp_table(ZP), Npix, Nb, Nh, PGT);
}

// ------------------------------------------------------------------------------------------------
// k_prep_pst : the state images from the float32 record SOL (one block per group).  The training step does not launch it:
// k_solve builds the images itself (qfa_step_kernels.h, template argument STATE); this kernel serves tools and tests.
// ------------------------------------------------------------------------------------------------
template <int KP>
__global__ __launch_bounds__(256) void k_prep_pst(const float *__restrict__ SOL, int B, int Nh, unsigned char *__restrict__ PST) {
    build_state<KP>(SOL + (size_t)16 * blockIdx.x * Cfg<KP>::NSOL, 16 * blockIdx.x, B, Nh, PST + (size_t)blockIdx.x * GTT<KP>::STATE_B,
                    threadIdx.x);
}

#ifndef QFA_GT_STAMPS
#define QFA_GT_STAMPS 0    // diagnostic build (tools/gt_stamps.sh): s_memtime shares of the walk of two waves of one workgroup
#endif
#if QFA_GT_STAMPS
__device__ unsigned long long qfa_gt_stamps[2 * 16];
#define GTS(i)                                                                                 \
    {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                     \
        unsigned long long t_;                                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");             \
        st_[i] += (unsigned)(t_ - st_last);                                                    \
        st_last = t_;                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                     \
    }
#else
#define GTS(i) {}
#endif
template <int KP, bool HASA, bool ZF, bool IDX>        // IDX: the batch carries row numbers (qfa_batch_t::rows)
__global__ __launch_bounds__(512, 2) void k_grads_t(qfa_params_t p, qfa_batch_t bt, qfa_tau_t tau, int B, int Npix, int Nb,
                                                    int Nh, GtPlan gp, const unsigned char *__restrict__ PGT,
                                                    const unsigned char *__restrict__ PST, const float4 *__restrict__ ZS,
                                                    float *__restrict__ accum,
                                                    float *__restrict__ slab, double *__restrict__ slabS, int slab_stride,
                                                    Scal64 *__restrict__ sc64) {
    using GT = GTT<KP>;
    __shared__ __attribute__((aligned(16))) unsigned char lds[GT::L_TOTAL];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv8 = wave_uniform(tid >> 6);
    const int lo = lane & 15, g = lane >> 4;
    const int item = ((int)blockIdx.x % gp.nxcd) * gp.per_xcd() + ((int)blockIdx.x / gp.nxcd);
    const bool idle = item >= gp.PB * gp.R;                 // (padding of the grid)
    const int rr = idle ? 0 : item / gp.PB, pb = idle ? 0 : item % gp.PB;
    const int G = (B + 15) >> 4;
    const int g0 = rr * gp.gpr;                             // first group of the range
    const int n = idle ? 0 : max(0, min(gp.gpr, G - g0));   // groups in the range (the same for every wave: the barriers)
    constexpr int TPW = GT::TPW, PXW = GT::PXW;
    const int wt = pb + gp.PB * wv8;                        // this wave's tile of PXW pixels
#ifndef QFA_GT_ONLY
#define QFA_GT_ONLY 0      // timing experiments (wrong results): 1 = waves 0..3 only, 2 = waves 4..7 only
#endif
    const bool active = !idle && wt < gp.T16 && (QFA_GT_ONLY == 0 || (QFA_GT_ONLY == 1) == (wv8 < 4));      // wave-uniform
    // 16-pixel tile j of the wave: column lo <-> pixel PXW wt + TPW lo + j
    int px[TPW];
    bool inb[TPW], blue[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        px[j] = PXW * wt + TPW * lo + j;
        inb[j] = active && px[j] < Npix;
        blue[j] = active && px[j] < Nb;
    }
    const bool blueTile = active && PXW * wt < Nb;          // wave-uniform: the wave tile has blue pixels
    const DevConsts k = load_consts(p, tau);
    const bool det = slab != nullptr;
    float *accF = det ? slab + (size_t)rr * (size_t)slab_stride : accum;
    float *accA = accF + (size_t)Npix * Nh;                 // sumA | gPsi | gOmega | cnt
    float *accS = accum + (size_t)Npix * Nh + 3 * (size_t)Npix + Nb;

#if QFA_GT_STAMPS
    unsigned st_[8];
    for (int i = 0; i < 8; ++i) st_[i] = 0;
    unsigned long long st_last;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(st_last)::"memory");
#endif
    // ---- static operands of the wave's tiles
    const unsigned char *tile[TPW];
    u32x4 IBh[TPW][GT::NKQ], IBm[TPW][GT::NKQ], IBl[TPW][GT::NKQ];
    float Psi[TPW], om[TPW], ti[TPW], pwi[TPW], l2i[TPW], offl[TPW];
    float it2[TPW], it1[TPW];                               // F16S1: the inverse powers of two of the lane's pixel (pairs, F)
    float sbeta[TPW];                                       // F16S3: the power of two of the lane's pixel for beta (<= 2^12 / max beta)
    f32x4 zfac = {1.f, 1.f, 1.f, 1.f};                      // F16S3: 2^(7 - zk) of the lane's four spectra of the group in stage 1 / 2
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        tile[j] = PGT + (size_t)(active ? TPW * wt + j : 0) * GT::TILE_B;
#pragma unroll
        for (int ks = 0; ks < GT::NKQ; ++ks) {
            IBh[j][ks] = *reinterpret_cast<const u32x4 *>(tile[j] + ks * GT::BLK_B + lane * 16);
            IBm[j][ks] = *reinterpret_cast<const u32x4 *>(tile[j] + ks * GT::BLK_B + 1024 + lane * 16);
            IBl[j][ks] = GT::F16S1 ? u32x4{0u, 0u, 0u, 0u} : *reinterpret_cast<const u32x4 *>(tile[j] + ks * GT::BLK_B + (GT::S1NP - 1) * 1024 + lane * 16);
        }
        const float *par = reinterpret_cast<const float *>(tile[j] + GT::OFF_PAR);
        Psi[j] = par[lo]; om[j] = par[16 + lo];
        it2[j] = GT::F16S1 ? par[GT::PAR_IT2 + lo] : 1.f; it1[j] = GT::F16S1 ? par[GT::PAR_IT1 + lo] : 1.f;
        sbeta[j] = GT::F16S3 ? par[GT::PAR_SBETA + lo] : 1.f;
        ti[j] = ZF ? par[32 + lo] : 0.f; pwi[j] = ZF ? par[48 + lo] : 0.f; l2i[j] = ZF ? par[64 + lo] : 0.f;
        // factored-z form: a red pixel of the tile that holds the boundary has omega = ti = pwi = 0 in its image
        // (prep_pgt_body) and offset 0 here: A = exp2(0) = 1 and omega zd = 0 come out of the blue arithmetic without a select
        offl[j] = blue[j] ? k.offp : 0.f;
    }

    // Every load of this prologue is consumed HERE: hipcc puts the s_waitcnt of a load in front of its first use, and a first
    // use inside the walk leaves a vmcnt(0..4) in the loop that drains the untracked DMA queue in every step (stage 2 of
    // a blue group took 2 550 cycles that way, tools/gt_stamps.sh)
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
#pragma unroll
        for (int ks = 0; ks < GT::NKQ; ++ks) {
            asm volatile("" ::"v"(IBh[j][ks]), "v"(IBm[j][ks]));
            if (!GT::F16S1) asm volatile("" ::"v"(IBl[j][ks]));
        }
        asm volatile("" ::"v"(Psi[j]), "v"(om[j]), "v"(ti[j]), "v"(pwi[j]), "v"(l2i[j]), "v"(it2[j]), "v"(it1[j]), "v"(sbeta[j]));
    }
    asm volatile("" ::"v"(k.tau0), "v"(k.c0), "v"(k.beta), "v"(k.t_amp), "v"(k.t_lscale), "v"(k.t_expo), "v"(k.t_off), "v"(k.offp),
                 "v"(k.k1), "v"(k.omc0));

    // ---- running sums
    f32x4 W[TPW][GT::NWT], gacc[TPW];
    float gPsi[TPW], gOm[TPW], sA[TPW];
    int cnt[TPW];
    f32x4 betaR[TPW], gamR[TPW];                            // stage 2 -> stage 3 (across one barrier)
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
#pragma unroll
        for (int a = 0; a < GT::NWT; ++a) W[j][a] = f32x4{0.f, 0.f, 0.f, 0.f};
        gacc[j] = betaR[j] = gamR[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        gPsi[j] = gOm[j] = sA[j] = 0.f;
        cnt[j] = 0;
    }
    double d_tau0 = 0.0, d_c0 = 0.0, d_beta = 0.0;

    // ---- spectra staging of this wave: per array [16 slots][PXW px] float (slot q holds row q ^ ((q >> 2) & 1): the rows
    // 4 g + r that a lane's reads touch then alternate between the two halves of the banks), mask bytes as PXW / 16 halves
    // [16 slots][16]
    unsigned char *stg = lds + GT::L_STG + wv8 * 2 * GT::STG_B;
    const bool fastp = active && PXW * wt + PXW - 1 < Npix;
    const bool zblue = !ZF && blueTile;                     // the tile stages zabs
    const bool fastz = !zblue || PXW * wt + PXW - 1 < Nb;
    const bool slow = active && !fastp;                     // the ragged last tile: 4-byte pieces
    const bool zstrad = active && fastp && !fastz;          // zabs form, the tile across the end of the blue side: zabs as 4-byte pieces
    const bool zfb = ZF && blueTile;                        // the tile stages the per-spectrum factors of the factored-z form
    // (no counted waits any more, round 5: every wave waits for all of its requests in front of the step's barrier)
    // first byte of the 4-byte mask piece (half h of the tile, piece pc) in its row; the ragged tile clamps it to Npix - 4
    auto mask_start = [&](int h, int pc) __attribute__((always_inline)) {
        const int st = PXW * wt + 16 * h + 4 * pc;
        return slow ? max(0, min(st, Npix - 4)) : st;
    };
    // Group t of the range: spectra s0 .. s0 + 15 of the batch.  Spectrum s is row batch_row(s) of the batch arrays (ABI v3:
    // rows / row_stride, qfa_common.h): every request takes a per-lane 64-bit address.  Indexed form: the row indices of a
    // group are lane-varying values that must be in registers when its requests are formed, two steps before its data are
    // used -- they ride along: the requests of group t carry the 16 indices of group t + 2 into STG_ROWS of the same
    // staging buffer (one more request per group, on the counted path), and stage_spectra(t + 2), which re-uses that
    // buffer, reads them there before its first request.  The first two groups (FIRST) read theirs from global memory.
    const unsigned RS = (unsigned)bt.row_stride;            // (elements; < 2^31: check_batch)
    constexpr bool idx = IDX;
    // row of the batch arrays <- row `row` of group t, whose indices wait in staging buffer bufi (FIRST: in global memory)
    auto row_index = [&](int t, int bufi, unsigned row, auto first_tag) __attribute__((always_inline)) -> unsigned {
        constexpr bool FIRST = decltype(first_tag)::value;
        const unsigned s0 = 16u * (unsigned)(g0 + t);
        if (!idx) return s0 + row;
        if (FIRST || QFA_TRACKED_LOADS) return (unsigned)bt.rows[s0 + row];
        return (unsigned)reinterpret_cast<const int *>(stg + bufi * GT::STG_B + GT::STG_ROWS)[row];
    };
    // the rows the lane's requests of a whole tile take (the ragged and the straddling tile read theirs where they need them):
    // r[i] = row of the 16-byte pieces of request i, m = row of the lane's mask piece.  Read by the caller AHEAD of
    // stage_spectra, with the LDS reads of the stage it sits in (a read right in front of the requests is ~100 cycles in the open)
    struct RowIdx {
        unsigned r[TPW], m;
    };
    auto read_rows = [&](int t, int bufi, auto first_tag) __attribute__((always_inline)) {
        RowIdx ri;
        const int last_row = min(15, B - 1 - 16 * (g0 + t));
        auto slot_row = [&](int q) __attribute__((always_inline)) { return (unsigned)min(q ^ ((q >> 2) & 1), last_row); };
        ri.m = row_index(t, bufi, slot_row(lane >> 2), first_tag);
        if constexpr (TPW == 1) ri.r[0] = ri.m;
        else {
#pragma unroll
            for (int i = 0; i < TPW; ++i) ri.r[i] = row_index(t, bufi, slot_row((64 / (4 * TPW)) * i + lane / (4 * TPW)), first_tag);
        }
        return ri;
    };
    // ---- round 5: the steady state of the staging requests without per-step address arithmetic.  A request of group t takes
    // array base + (16 (g0 + t) + slot row) x row stride + the pixel of the lane's piece: in batch order (no row table), for a
    // tile inside the pixel axis (zabs form: inside the blue side or the red one) and a FULL group of 16 spectra everything but
    // 16 (g0 + t) x row stride is a per-lane constant of the launch.  That part lives in wave-uniform 64-bit pointers (SGPR pairs)
    // which stage2 advances by 16 rows per call; the lane's offsets are three VGPRs formed once.  (The form it replaces rebuilt
    // bases and offsets in every step: ~45 scalar instructions, a reload of a kernel argument and its wait -- 620 cycles per
    // group and wave in the stamps; a scalar instruction outside an MFMA's shadow costs the wave 4 cycles like any other,
    // tools/ubench/issue_mix.hip.)  The first two groups, the partial last group of a batch, ragged / straddling tiles, two
    // tiles per wave and the indexed form keep the general path below.
#ifndef QFA_GT_RUNPTR
#define QFA_GT_RUNPTR 1
#endif
    constexpr bool RUNP = QFA_GT_RUNPTR && !IDX && TPW == 1 && !QFA_TRACKED_LOADS;
    struct SpecRun {
        const unsigned char *d, *e, *z, *m;       // group t + 2 of the next stage2 call; e, z, m biased by the LDS offsets of their arrays
    } sr{nullptr, nullptr, nullptr, nullptr};
    unsigned sr_vo = 0u, sr_vz = 0u, sr_om = 0u;  // byte offsets of the lane's pieces: delta / sigma, zabs (or the factors ZS), mask
    const bool runok = RUNP && active && !slow && !zstrad;
    if (RUNP) {
        const size_t r2 = (size_t)16 * (size_t)(g0 + 2);
        const unsigned q = (unsigned)lane >> 2, row = q ^ ((q >> 2) & 1u), pc = 16u * (unsigned)wt + 4u * ((unsigned)lane & 3u);
        sr.d = uniform_ptr(reinterpret_cast<const unsigned char *>(bt.delta + r2 * RS));
        sr.e = uniform_ptr(reinterpret_cast<const unsigned char *>(bt.error + r2 * RS) - GT::STG_ARR);
        sr.m = uniform_ptr(reinterpret_cast<const unsigned char *>(bt.mask + r2 * RS) - GT::STG_MASK);
        sr.z = zblue ? uniform_ptr(reinterpret_cast<const unsigned char *>(bt.zabs + r2 * (size_t)Nb) - 2 * GT::STG_ARR)
                     : (zfb ? uniform_ptr(reinterpret_cast<const unsigned char *>(ZS + r2) - 2 * GT::STG_ARR) : sr.d);
        sr_om = row * RS + pc;
        sr_vo = 4u * sr_om;
        sr_vz = zblue ? 4u * (row * (unsigned)Nb + pc) : 16u * ((unsigned)lane & 15u);
    }
    // the requests of one group from the running pointers (factored-z form: the 16 float4 factors as a request of all 64
    // lanes, lanes 16.. repeat them -- the 1 KiB they fill is the slot the zabs tile has in the other form)
    // Request k of the group the running pointers stand at: 0 delta (it writes M0 for all four), 1 sigma, 2 zabs / the factors,
    // 3 the masks.  Round 5: they are issued ONE BY ONE between the MFMA groups of stage 3 -- back to back in stage 2 each waited
    // until the texture path had taken the one before (4 requests: 625 cycles of the blue waves' 5 400 per group in the stamps,
    // whatever the number of scalar instructions around them), between MFMAs they cost nothing (the state pieces never did).
    auto stage_req = [&](int bufi, int k) __attribute__((always_inline)) {
        if (k == 0) {
            const unsigned dst = wave_uniform(lds_addr(stg + bufi * GT::STG_B));
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(sr_vo), "s"(sr.d), "s"(dst) : "memory");
        } else if (k == 1) asm volatile("global_load_lds_dwordx4 %0, %1 offset:1024" ::"v"(sr_vo), "s"(sr.e) : "memory");
        else if (k == 2) asm volatile("global_load_lds_dwordx4 %0, %1 offset:2048" ::"v"(sr_vz), "s"(sr.z) : "memory");
        else asm volatile("global_load_lds_dword %0, %1 offset:3072" ::"v"(sr_om), "s"(sr.m) : "memory");
    };
    const size_t sr_zinc = zblue ? (size_t)64 * (size_t)(unsigned)Nb : (zfb ? (size_t)256 : (size_t)0);
    auto advance_run = [&]() __attribute__((always_inline)) {
        if (!RUNP) return;
#ifdef QFA_GT_ABL_L2
        return;                // timing experiment (wrong results): every group re-reads the rows of group 2 -- the spectra come out of L2
#endif
        sr.d += (size_t)64 * RS;
        sr.e += (size_t)64 * RS;
        sr.m += (size_t)16 * RS;
        sr.z += sr_zinc;
    };
    auto stage_spectra = [&](int t, int bufi, const RowIdx &ri, auto first_tag) __attribute__((always_inline)) {
        if (!active) return;
        const int s0 = 16 * (g0 + t);
        const int last_row = min(15, B - 1 - s0);                                     // wave-uniform, >= 0
        const unsigned dst = wave_uniform(lds_addr(stg + bufi * GT::STG_B));
        auto slot_row = [&](int q) __attribute__((always_inline)) { return (unsigned)min(q ^ ((q >> 2) & 1), last_row); };
        auto Rof = [&](unsigned row) __attribute__((always_inline)) -> unsigned { return row_index(t, bufi, row, first_tag); };
#if QFA_TRACKED_LOADS
        {
            float *sf = reinterpret_cast<float *>(stg + bufi * GT::STG_B);
            unsigned char *mb = stg + bufi * GT::STG_B + GT::STG_MASK;
#pragma unroll
            for (int i = 0; i < 4 * TPW; ++i) {
                const int e = 64 * i + lane, q = e / PXW, pl = e % PXW;               // (slot, pixel of the wave tile)
                const unsigned R = Rof(slot_row(q));
                const int pxx = PXW * wt + pl;
                const unsigned long long o = (unsigned long long)R * RS + (unsigned)min(pxx, Npix - 1);
                sf[0 * (GT::STG_ARR / 4) + q * PXW + pl] = bt.delta[o];
                sf[1 * (GT::STG_ARR / 4) + q * PXW + pl] = bt.error[o];
                if (zblue) sf[2 * (GT::STG_ARR / 4) + q * PXW + pl] = bt.zabs[(unsigned long long)R * (unsigned)Nb + (unsigned)min(pxx, Nb - 1)];
                mb[(pl >> 4) * 256 + q * 16 + (pl & 15)] = pxx < Npix ? bt.mask[o] : (unsigned char)0;
            }
            if (zfb && lane < 16) reinterpret_cast<float4 *>(sf + 2 * (GT::STG_ARR / 4))[lane] = ZS[s0 + min(lane, last_row)];
            (void)dst;
            return;
        }
#endif
        [&]() __attribute__((always_inline)) {
        // the masks of the forms that do not carry them in their one statement: per half of 16 pixels one request of 4-byte
        // pieces (lane = (slot, piece)); their row index first, with the other LDS reads of this call
        const unsigned Rm = ri.m;
        if constexpr (!IDX) {
            // batch order = storage order: the 16 rows of a group are neighbours -- wave-uniform bases (the group's first row,
            // SGPRs) and 32-bit per-lane offsets, no address arithmetic beyond one multiply-add per request (15 RS < 2^29: check_batch)
            if (!slow && !zstrad) {
                const float *dbase = uniform_ptr(bt.delta + (size_t)s0 * RS);
                const float *ebase = uniform_ptr(bt.error + (size_t)s0 * RS);
                const uint8_t *mbase = uniform_ptr(bt.mask + (size_t)s0 * RS);
                const float *zbase = zblue ? uniform_ptr(bt.zabs + (size_t)s0 * Nb) : dbase;
                if (zfb && lane < 16) glds16a(uniform_ptr(ZS + s0), 16u * (unsigned)min(lane, last_row), dst + 2 * GT::STG_ARR);
                if constexpr (TPW == 1) {
                    const unsigned row = slot_row(lane >> 2);
                    const unsigned pc = 16u * (unsigned)wt + 4u * (unsigned)(lane & 3);
                    const unsigned o = row * RS + pc;
                    const unsigned vo = 4u * o, vz = 4u * (row * (unsigned)Nb + pc);
                    const unsigned char *eb = reinterpret_cast<const unsigned char *>(ebase) - GT::STG_ARR;
                    const unsigned char *zb = reinterpret_cast<const unsigned char *>(zbase) - 2 * GT::STG_ARR;
                    const unsigned char *mb_ = reinterpret_cast<const unsigned char *>(mbase) - GT::STG_MASK;
                    if (zblue)
                        asm volatile("s_mov_b32 m0, %7\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %3\n\t"
                                     "global_load_lds_dwordx4 %0, %4 offset:1024\n\tglobal_load_lds_dwordx4 %1, %5 offset:2048\n\t"
                                     "global_load_lds_dword %2, %6 offset:3072"
                                     ::"v"(vo), "v"(vz), "v"(o), "s"(dbase), "s"(eb), "s"(zb), "s"(mb_), "s"(dst) : "memory");
                    else
                        asm volatile("s_mov_b32 m0, %5\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %2\n\t"
                                     "global_load_lds_dwordx4 %0, %3 offset:1024\n\tglobal_load_lds_dword %1, %4 offset:3072"
                                     ::"v"(vo), "v"(o), "s"(dbase), "s"(eb), "s"(mb_), "s"(dst) : "memory");
                } else {
                    constexpr int LPR = 4 * TPW, SPI = 64 / LPR;
#pragma unroll
                    for (int h = 0; h < TPW; ++h)
                        glds4a(mbase, slot_row(lane >> 2) * RS + (unsigned)mask_start(h, lane & 3), dst + GT::STG_MASK + h * 256);
#pragma unroll
                    for (int i = 0; i < TPW; ++i) {
                        const unsigned row = slot_row(SPI * i + lane / LPR);
                        const unsigned pc = (unsigned)(PXW * wt) + 4u * (unsigned)(lane % LPR);
                        glds16a(dbase, 4u * (row * RS + pc), dst + i * 1024);
                        glds16a(ebase, 4u * (row * RS + pc), dst + GT::STG_ARR + i * 1024);
                        if (zblue) glds16a(zbase, 4u * (row * (unsigned)Nb + pc), dst + 2 * GT::STG_ARR + i * 1024);
                    }
                }
                return;
            }
        }
        if (!slow) {
            if constexpr (TPW == 1) {
                const unsigned R = ri.r[0];                                           // staging slot lane >> 2
                unsigned pc = 16u * (unsigned)wt + 4u * (unsigned)(lane & 3);         // first pixel of the lane's piece
                asm volatile("" : "+v"(pc));          // (opaque: base + 4 pc per array would otherwise live in registers for the whole walk)
                const unsigned long long o = (unsigned long long)R * RS + pc;
                // factored-z form: the factors of the 16 spectra (row r at float4 index r of array 2), one request of 16 lanes
                if (zfb && lane < 16) glds16a(uniform_ptr(ZS + s0), 16u * (unsigned)min(lane, last_row), dst + 2 * GT::STG_ARR);
                if (zstrad) {                   // zabs of the straddling tile: clamped pixels, 4 slots per request (one wave: rolled)
#pragma unroll 1
                    for (int i = 0; i < 4; ++i) {
                        int pz = min(16 * wt + lo, Nb - 1);
                        asm volatile("" : "+v"(pz));
                        glds4p(lane_ptr<2>(bt.zabs, (unsigned long long)Rof(slot_row(4 * i + g)) * (unsigned)Nb + (unsigned)pz), dst + 2 * GT::STG_ARR + i * 256);
                    }
                }
                // delta, sigma, (zabs,) the masks (16 rows x 16 bytes as 4-byte pieces): one write of M0, the LDS offsets in the
                // instructions' immediate fields, taken off the global addresses again
                const unsigned char *pd = lane_ptr<2>(bt.delta, o);
                const unsigned char *pe = lane_ptr<2>(reinterpret_cast<const unsigned char *>(bt.error) - GT::STG_ARR, o);
                const unsigned char *pm = lane_ptr<0>(bt.mask - GT::STG_MASK, o);
                if (zblue && !zstrad) {
                    const unsigned char *pz = lane_ptr<2>(reinterpret_cast<const unsigned char *>(bt.zabs) - 2 * GT::STG_ARR, (unsigned long long)R * (unsigned)Nb + pc);
                    asm volatile("s_mov_b32 m0, %4\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
                                 "global_load_lds_dwordx4 %1, off offset:1024\n\tglobal_load_lds_dwordx4 %2, off offset:2048\n\t"
                                 "global_load_lds_dword %3, off offset:3072"
                                 ::"v"(pd), "v"(pe), "v"(pz), "v"(pm), "s"(dst) : "memory");
                } else
                    asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\t"
                                 "global_load_lds_dwordx4 %1, off offset:1024\n\tglobal_load_lds_dword %2, off offset:3072"
                                 ::"v"(pd), "v"(pe), "v"(pm), "s"(dst) : "memory");
            } else {
                // 16-byte pieces: a row of the wave tile is 4 TPW lanes, a request 16 / TPW slots
                constexpr int LPR = 4 * TPW;
                if (zfb && lane < 16) glds16a(uniform_ptr(ZS + s0), 16u * (unsigned)min(lane, last_row), dst + 2 * GT::STG_ARR);
#pragma unroll
                for (int h = 0; h < TPW; ++h) {
                    unsigned ms = (unsigned)mask_start(h, lane & 3);
                    asm volatile("" : "+v"(ms));
                    glds4p(lane_ptr<0>(bt.mask, (unsigned long long)Rm * RS + ms), dst + GT::STG_MASK + h * 256);
                }
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    unsigned pc = (unsigned)(PXW * wt) + 4u * (unsigned)(lane % LPR);
                    asm volatile("" : "+v"(pc));      // (opaque: base + 4 pc per array would otherwise live in registers for the whole walk)
                    const unsigned long long o = (unsigned long long)ri.r[i] * RS + pc;
                    glds16p(lane_ptr<2>(bt.delta, o), dst + i * 1024);
                    glds16p(lane_ptr<2>(bt.error, o), dst + GT::STG_ARR + i * 1024);
                    if (zblue && !zstrad) glds16p(lane_ptr<2>(bt.zabs, (unsigned long long)ri.r[i] * (unsigned)Nb + pc), dst + 2 * GT::STG_ARR + i * 1024);
                }
                if (zstrad) {                   // (one wave of the launch: rolled)
#pragma unroll 1
                    for (int i = 0; i < 4 * TPW; ++i) {
                        const int e = 64 * i + lane;
                        int pz = min(PXW * wt + e % PXW, Nb - 1);
                        asm volatile("" : "+v"(pz));
                        glds4p(lane_ptr<2>(bt.zabs, (unsigned long long)Rof(slot_row(e / PXW)) * (unsigned)Nb + (unsigned)pz), dst + 2 * GT::STG_ARR + i * 256);
                    }
                }
            }
            return;
        }
        // ragged tile (the last one of a pixel axis that is no multiple of PXW): 4-byte pieces with the pixel clamped per lane;
        // the masks as 4-byte pieces whose start is clamped to Npix - 4 (mask_pos below is where the lane then finds
        // its byte) -- everything on the counted path.  (One wave of the launch: rolled loops, nothing kept in registers.)
        if (zfb && lane < 16) glds16a(uniform_ptr(ZS + s0), 16u * (unsigned)min(lane, last_row), dst + 2 * GT::STG_ARR);
#pragma unroll 1
        for (int h = 0; h < TPW; ++h) {
            unsigned ms = (unsigned)mask_start(h, lane & 3);
            asm volatile("" : "+v"(ms));
            glds4p(lane_ptr<0>(bt.mask, (unsigned long long)Rm * RS + ms), dst + GT::STG_MASK + h * 256);
        }
#pragma unroll 1
        for (int i = 0; i < 4 * TPW; ++i) {
            const int e = 64 * i + lane, pl = e % PXW;
            const unsigned R = Rof(slot_row(e / PXW));
            int pxx = PXW * wt + pl;
            asm volatile("" : "+v"(pxx));
            const unsigned long long o = (unsigned long long)R * RS + (unsigned)min(pxx, Npix - 1);
            glds4p(lane_ptr<2>(bt.delta, o), dst + i * 256);
            glds4p(lane_ptr<2>(bt.error, o), dst + GT::STG_ARR + i * 256);
            if (zblue) glds4p(lane_ptr<2>(bt.zabs, (unsigned long long)R * (unsigned)Nb + (unsigned)min(pxx, Nb - 1)), dst + 2 * GT::STG_ARR + i * 256);
        }
        }();
        // indexed form: the row indices of group t + 2 (clamped to the range's last group: the request count stays fixed)
        // behind the requests whose addresses were formed from this buffer's previous indices
        if (idx) {
            const int t2 = min(t + 2, n - 1), s2 = 16 * (g0 + t2);
            if (lane < 16) glds4a(uniform_ptr(bt.rows + s2), 4u * (unsigned)min(lane, B - 1 - s2), dst + GT::STG_ROWS);
        }
    };
    // where the mask byte of the lane's pixel of tile j sits inside a slot's staged bytes: half, then byte of the half
    int mask_pos[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) {
        const int pl = TPW * lo + j, h = pl >> 4, pi = pl & 15;
        mask_pos[j] = h * 256 + ((slow && !QFA_TRACKED_LOADS) ? 4 * (pi >> 2) + min(3, PXW * wt + pl - mask_start(h, pi >> 2)) : pi);
    }

    // ---- the state parts: a contiguous run of 1-KiB pieces per wave.  A request costs the issuing wave ~80 - 100 cycles, and
    // the waves with a blue tile are the critical path of a step (stage 2 of a blue group is 1 200 cycles against 800): the
    // pieces go to the waves with a red tile or none ("duty" waves) when there are at least five of them, else to all eight
#ifndef QFA_GT_DUTY
#define QFA_GT_DUTY 1
#endif
    int nduty = 0, drank = 0;
    bool duty = true;
    {
        int nd = 0, rk = 0;
#pragma unroll
        for (int w = 0; w < GT::NW; ++w) {
            const int tw = pb + gp.PB * w;
            const bool d = !(tw < gp.T16 && PXW * tw < Nb);
            if (w < wv8 && d) ++rk;
            if (d) ++nd;
        }
        if (QFA_GT_DUTY && nd >= 5) { nduty = nd; drank = rk; duty = !blueTile; }
        else { nduty = GT::NW; drank = wv8; duty = true; }
    }
    const int s1_q = GT::S1_PCS / nduty, s1_r = GT::S1_PCS % nduty, z_q = GT::Z_PCS / nduty, z_r = GT::Z_PCS % nduty;
    const int s1_first = drank * s1_q + min(drank, s1_r), z_first = drank * z_q + min(drank, z_r);
    const int s1_req = duty ? s1_q + (drank < s1_r ? 1 : 0) : 0, z_req = duty ? z_q + (drank < z_r ? 1 : 0) : 0;     // <= 4, <= 7
    auto issue_S1 = [&](int t) __attribute__((always_inline)) {
        const unsigned char *src = uniform_ptr(PST + (size_t)(g0 + t) * GT::STATE_B + s1_first * 1024);
        const unsigned dst = wave_uniform(lds_addr(lds + GT::L_S1 + (t & 1) * GT::S1P_B + s1_first * 1024));
        for (int j = 0; j < s1_req; ++j) glds16a(src + 1024 * j, (unsigned)lane * 16u, dst + 1024 * j);
    };
    auto issue_Z = [&](int t) __attribute__((always_inline)) {
        const unsigned char *src = uniform_ptr(PST + (size_t)(g0 + t) * GT::STATE_B + GT::S1P_B + z_first * 1024);
        const unsigned dst = wave_uniform(lds_addr(lds + GT::L_Z + (t & 1) * GT::ZP_B + z_first * 1024));
        for (int j = 0; j < z_req; ++j) glds16a(src + 1024 * j, (unsigned)lane * 16u, dst + 1024 * j);
    };

    // A part's pieces are issued one by one BETWEEN the MFMA groups of the stage that runs in the same half-step: eight waves
    // that issue their runs together behind the barrier fill the texture path's queue (16 cycles per piece) and wait in
    // front of it -- 900 - 1 400 cycles per group and wave (tools/gt_stamps.sh).
    // Round 5: a piece is ONE instruction.  The wave's run of a part is contiguous in global memory and in LDS, so the stage
    // writes M0 once (part_begin: LDS address of the run + 4096) and piece j moves both addresses by its immediate offset
    // 1024 j - 4096 (signed 13 bits: eight pieces); the source pointers run along the walk (+ STATE_B per group, biased by
    // 4096 as well) instead of being rebuilt from (g0 + t) per part, and the waves WITHOUT duty (the blue ones when there are
    // five others) run instantiations of the walk that contain no piece code at all (NoPart) -- they used to execute a
    // compare and a branch at each of the 22 piece sites of a step.
    // (M0 belongs to these statements: nothing else in the walk may write it between part_begin and the stage's last piece --
    // stage2 calls part_begin behind its spectra requests; tools/audit_asm_loads.py refuses a compiler-made write of M0.)
    struct Part {
        const unsigned char *src;      // biased by + 4096 (tracked build: unbiased)
        unsigned dst;                  // LDS byte address of the run, biased the same way
        int cnt;
    };
    struct NoPart {};
    constexpr unsigned PBIAS = QFA_TRACKED_LOADS ? 0u : 4096u;
    const unsigned lane16 = (unsigned)lane * 16u;
    auto part_begin = [&](const auto &pt) __attribute__((always_inline)) {
        if constexpr (std::is_same_v<std::decay_t<decltype(pt)>, Part>) {
#if !QFA_TRACKED_LOADS
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(pt.dst));
#endif
        }
    };
    auto piece = [&](const auto &pt, int j) __attribute__((always_inline)) {
        if constexpr (std::is_same_v<std::decay_t<decltype(pt)>, Part>) {
            if (j < pt.cnt) {
#if QFA_TRACKED_LOADS
                glds16a(pt.src + 1024 * j, lane16, pt.dst + 1024 * j);
#else
#define QFA_PC(J) case J: asm volatile("global_load_lds_dwordx4 %0, %1 offset:%2" ::"v"(lane16), "s"(pt.src), "n"(1024 * J - 4096)); break;
                switch (j) { QFA_PC(0) QFA_PC(1) QFA_PC(2) QFA_PC(3) QFA_PC(4) QFA_PC(5) QFA_PC(6) QFA_PC(7) default: break; }
#undef QFA_PC
#endif
            }
        }
    };
    static_assert((GT::Z_PCS + 4) / 5 <= 8 && (GT::S1_PCS + 4) / 5 <= 8, "a wave's run of a part: at most eight pieces (immediate offsets)");

    // ---- stage 1 of group t (its results wait in afy / aq for stage 2, one half-step later); the operands of a K block are
    // read once for the wave's TPW tiles
    f32x4 afy[TPW], aq[TPW];
#pragma unroll
    for (int j = 0; j < TPW; ++j) afy[j] = aq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto stage1 = [&](int t, const auto &pt) __attribute__((always_inline)) {
        if constexpr (QFA_GT_SETPRIO != 0 && (QFA_GT_PRIO_STAGES & 1) != 0) __builtin_amdgcn_s_setprio(QFA_GT_SETPRIO);
        part_begin(pt);
        const unsigned char *sp = lds + GT::L_S1 + (t & 1) * GT::S1P_B + lane * 16;
#pragma unroll
        for (int j = 0; j < TPW; ++j) afy[j] = aq[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        constexpr int NP = GT::S1NP;
        u32x4 aop[2][NP];
#pragma unroll
        for (int pc = 0; pc < NP; ++pc) aop[0][pc] = *reinterpret_cast<const u32x4 *>(sp + pc * 1024);
        // F16S1: the inverse powers of two of the lane's four spectra (Cinv', y), out of the y block's free K slots
        f32x4 isc[2] = {f32x4{1.f, 1.f, 1.f, 1.f}, f32x4{1.f, 1.f, 1.f, 1.f}};
        if constexpr (GT::F16S1) {
            const unsigned char *sq = lds + GT::L_S1 + (t & 1) * GT::S1P_B + GT::S1_SCALES + 32 * g;
            isc[0] = *reinterpret_cast<const f32x4 *>(sq);          // 1 / scale(Cinv'), 1 / scale(y) of spectra 4 g, 4 g + 1
            isc[1] = *reinterpret_cast<const f32x4 *>(sq + 16);     // ... of 4 g + 2, 4 g + 3
            if constexpr (GT::F16S3)
                zfac = *reinterpret_cast<const f32x4 *>(lds + GT::L_S1 + (t & 1) * GT::S1P_B + GT::S1_ZFAC + 16 * g);
        }
#pragma unroll
        for (int ks = 0; ks < GT::NKS; ++ks) {
            if (ks + 1 < GT::NKS) {
#pragma unroll
                for (int pc = 0; pc < NP; ++pc)
                    aop[(ks + 1) & 1][pc] = *reinterpret_cast<const u32x4 *>(sp + (ks + 1) * GT::BLK_B + pc * 1024);
            }
            piece(pt, ks);
            if (ks == GT::NKS - 1) piece(pt, GT::NKS);             // (a seventh site: the Z part of the waves 4..7 rides here, QFA_GT_ZS1)
            const u32x4 &ah = aop[ks & 1][0], &am = aop[ks & 1][1], &al = aop[ks & 1][NP - 1];
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                if constexpr (GT::F16S1) {
                    if (ks < GT::NKQ) aq[j] = xdl3h(ah, am, IBh[j][ks], IBm[j][ks], aq[j]);
                    else afy[j] = xdl3h(ah, am, IBh[j][GT::NKQ - 1], IBm[j][GT::NKQ - 1], afy[j]);                          // the y block
                } else {
                    if (ks < GT::NKQ) aq[j] = xdl6(ah, am, al, IBh[j][ks], IBm[j][ks], IBl[j][ks], aq[j]);
                    else afy[j] = xdl6(ah, am, al, IBh[j][GT::NKQ - 1], IBm[j][GT::NKQ - 1], IBl[j][GT::NKQ - 1], afy[j]);     // the y block
                }
            }
        }
        if constexpr (GT::F16S1) {             // the powers of two back in: element r <-> spectrum 4 g + r, the lane's pixel
#pragma unroll
            for (int j = 0; j < TPW; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    aq[j][r] = (aq[j][r] * isc[r >> 1][2 * (r & 1)]) * it2[j];          // (one after the other: the PRODUCT of the
                    afy[j][r] = (afy[j][r] * isc[r >> 1][2 * (r & 1) + 1]) * it1[j];    // two powers may leave the float32 range)
                }
        }
        GTS(5)
        if constexpr (QFA_GT_SETPRIO != 0 && (QFA_GT_PRIO_STAGES & 1) != 0) __builtin_amdgcn_s_setprio(0);
    };
    // ---- the lane's 4 TPW elements of group t out of the staging buffer (spectra 4 g + r at the lane's pixels); stage 2 pins
    // sigma and the mask behind its wait: left to itself hipcc reads sigma under a branch on the mask, one LDS round trip after
    // the other (3 600 cycles for stage 2 of a blue group, tools/gt_stamps.sh).  (Volatile reads wait one by one: 2 200
    // cycles.  Issued a stage ahead, in front of the MFMAs of stage 1, they made stage 1 2 400 cycles long instead of 1 100.)
    float dv[TPW][4], sgv[TPW][4], zv[TPW][4];
    unsigned mk[TPW][4];
    RowIdx ri_next{};          // indexed form: the rows of group t + 2, read by stage 2 (t), used by stage 3 (t)
#pragma unroll
    for (int j = 0; j < TPW; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) { dv[j][r] = sgv[j][r] = zv[j][r] = 0.f; mk[j][r] = 0u; }
    auto take = [&](int t) __attribute__((always_inline)) {
        const unsigned char *sb = stg + (t & 1) * GT::STG_B;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int q = (4 * g + r) ^ (g & 1);
            if constexpr (TPW == 1) {
                dv[0][r] = *reinterpret_cast<const float *>(sb + (q * 16 + lo) * 4);
                sgv[0][r] = *reinterpret_cast<const float *>(sb + GT::STG_ARR + (q * 16 + lo) * 4);
                if (!ZF && blueTile) zv[0][r] = *reinterpret_cast<const float *>(sb + 2 * GT::STG_ARR + (q * 16 + lo) * 4);
            } else {                                       // the lane's two adjacent pixels: one 8-byte read per array
                typedef float f32x2t __attribute__((ext_vector_type(2)));
                const f32x2t d2 = *reinterpret_cast<const f32x2t *>(sb + (q * PXW + TPW * lo) * 4);
                const f32x2t s2 = *reinterpret_cast<const f32x2t *>(sb + GT::STG_ARR + (q * PXW + TPW * lo) * 4);
                dv[0][r] = d2[0]; dv[TPW - 1][r] = d2[1];
                sgv[0][r] = s2[0]; sgv[TPW - 1][r] = s2[1];
                if (!ZF && blueTile) {
                    const f32x2t z2 = *reinterpret_cast<const f32x2t *>(sb + 2 * GT::STG_ARR + (q * PXW + TPW * lo) * 4);
                    zv[0][r] = z2[0]; zv[TPW - 1][r] = z2[1];
                }
            }
#pragma unroll
            for (int j = 0; j < TPW; ++j) mk[j][r] = sb[GT::STG_MASK + q * 16 + mask_pos[j]];
        }
    };
    // ---- stage 2 of group t
    auto stage2_t = [&](auto blue_tag, int t, const auto &pt) __attribute__((always_inline)) {
        constexpr bool BLUE = decltype(blue_tag)::value;       // (the tile has blue pixels: wave-uniform, one branch per stage)
        const int s0 = 16 * (g0 + t);
        take(t);
        float zqx[4] = {0.f, 0.f, 0.f, 0.f}, zqy[4] = {0.f, 0.f, 0.f, 0.f}, zqz[4] = {0.f, 0.f, 0.f, 0.f};
        if (ZF && BLUE) {      // (the factors of the lane's four spectra)
            const unsigned char *sb = stg + (t & 1) * GT::STG_B;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 q = reinterpret_cast<const float4 *>(sb + 2 * GT::STG_ARR)[4 * g + r];
                zqx[r] = q.x; zqy[r] = q.y; zqz[r] = q.z;
            }
        }
        if (IDX && t + 2 < n) ri_next = read_rows(t + 2, t & 1, std::false_type{});      // (for stage 3's requests)
        bool wvm[TPW][4];                                        // element counts: mask & spectrum < B & pixel < Npix (lane masks)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // staging buffer read: it may be overwritten now
#pragma unroll
        for (int j = 0; j < TPW; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(sgv[j][r]), "+v"(mk[j][r]));
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < TPW; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) wvm[j][r] = inb[j] & (s0 + 4 * g + r < B) & (mk[j][r] != 0u);
        GTS(6)
        part_begin(pt);
        piece(pt, 0);
        piece(pt, 1);
        __builtin_amdgcn_sched_barrier(0);
        GTS(2)
        // gPsi / gOmega accumulate h = 2 dG (halved once, at the store); the walk counts valid elements in integers (one
        // add-with-carry each).  Factored-z form: pw = zqy[r] pwi[j] and l2 = zqz[r] + l2i[j] separate into the spectrum's and
        // the pixel's factor, so the tau0 / beta sums of :142-143 need e zqy[r] and e zqy[r] zqz[r] per element only; the
        // pixel's factors multiply the group's sums
        float t_tau0 = 0.f, t_c0 = 0.f, t_beta = 0.f;
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
        float e1 = 0.f, e2 = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool wv_ = wvm[j][r];
            const float dd = wv_ ? dv[j][r] : 0.f;
            const float sg = sgv[j][r];
            if (BLUE) {
                float l2 = 0.f, pw, Ab, re;
                if (ZF) {                                                                         // qfa_common.h, ZFac
                    pw = zqy[r] * pwi[j];
                    Ab = fast_exp2(fmaf(zqx[r], ti[j], offl[j]));                                 // QFA/model.py:125
                    re = k.omc0 - fast_exp2(k.k1 * pw);                                           // QFA/utils.py:91
                } else {
                    l2 = fast_log2(1.0f + zv[j][r]);
                    pw = fast_exp2(k.beta * l2);
                    const float tauv = k.t_amp * fast_exp2(k.t_expo * (l2 + k.t_lscale)) + k.t_off;   // QFA/utils.py:105-141
                    Ab = fast_exp2(-tauv * QFA_LOG2E);                                            // QFA/model.py:125
                    re = 1.0f - k.c0 - fast_exp2(-k.tau0 * pw * QFA_LOG2E);                       // QFA/utils.py:91
                }
                if (HASA) Ab = bt.A_blue[(size_t)min(s0 + 4 * g + r, B - 1) * Nb + (unsigned)min(px[j], Nb - 1)];   // custom tau callable
                const float Av = (ZF && !HASA) ? Ab : (blue[j] ? Ab : 1.f);
                const float zd = (ZF && !HASA) ? re * re : (blue[j] ? re * re : 0.f);
                const float A2 = Av * Av;
                const float ozd = om[j] * zd;
                const float D = A2 * Psi[j] + ozd + sg * sg;
                const float wD = wv_ ? fast_rcp(D) : 0.f;
                const float wDA = wD * Av;
                const float uu = wD * (dd - Av * afy[j][r]);                // (Sigma^-1 delta)_i
                const float dS = wD - wDA * wDA * aq[j][r];                 // diag(Sigma^-1)_i
                const float h = dS - uu * uu;                               // 2 dG, QFA/model.py:136,138
                gPsi[j] = fmaf(A2, h, gPsi[j]);                             // :139 (x 2)
                gOm[j] = fmaf(h, zd, gOm[j]);                               // :140 (x 2)
                const float root = fmaf(-k.tau0, pw, k.omc0);               // :141
                const float e = h * ozd * zd * root;                        // dG (omega zd) zd 2 root
                if (ZF) {
                    const float ep = e * zqy[r];
                    e1 += ep;                                               // :142 / pwi[j]
                    e2 = fmaf(ep, zqz[r], e2);                              // :143, the spectrum's part of l2
                } else {
                    t_tau0 -= e * pw;                                       // :142
                    t_beta -= e * (k.tau0 * pw * (l2 * QFA_LN2));           // :143
                }
                t_c0 -= e;                                                  // :144
                cnt[j] += wv_ ? 1 : 0;
                const float bb = wDA * Av;
                sA[j] += bb * Av;
                betaR[j][r] = GT::F16S3 ? bb * (sbeta[j] * zfac[r]) : bb;
                gamR[j][r] = Av * uu;
            } else {                                                        // red tile: A = 1, zd = 0
                const float D = Psi[j] + sg * sg;
                const float wD = wv_ ? fast_rcp(D) : 0.f;
                const float uu = wD * (dd - afy[j][r]);
                const float dS = wD - wD * wD * aq[j][r];
                gPsi[j] += dS - uu * uu;
                cnt[j] += wv_ ? 1 : 0;
                betaR[j][r] = GT::F16S3 ? wD * (sbeta[j] * zfac[r]) : wD;
                sA[j] += wD;
                gamR[j][r] = uu;
            }
            if (r & 1) {                                       // two elements at a time: bounds the live temporaries
                __builtin_amdgcn_sched_barrier(0);
                if (j == TPW - 1) {                            // (the state pieces of this stage: behind the wave's last pairs)
                    piece(pt, r == 1 ? 2 : 4);
                    piece(pt, r == 1 ? 3 : 5);
                    if (r == 3) piece(pt, 6);
                }
            }
        }
        if (BLUE && ZF) {
            const float t1 = pwi[j] * e1;
            t_tau0 -= t1;
            t_beta -= k.tau0 * QFA_LN2 * fmaf(pwi[j], e2, l2i[j] * t1);
        }
        }
        if (BLUE) {                                            // float32 inside a group, float64 across the walk
            d_tau0 += (double)t_tau0;
            d_c0 += (double)t_c0;
            d_beta += (double)t_beta;
        }
    };
    auto stage2 = [&](int t, const auto &pt) __attribute__((always_inline)) {       // (the prologue's call)
        if (blueTile) stage2_t(std::true_type{}, t, pt);
        else stage2_t(std::false_type{}, t, pt);
    };

    // ---- stage 3 of group t: W[a] += Z pieces x the lane's own beta pieces (K = spectrum), the gamma term likewise; the Z
    // operands of a column tile are read once for the wave's TPW tiles
    // The staging requests of group t + 2 (buffer t & 1, read for the last time by stage 2 (t)) are issued here: the running-pointer
    // form one request per column tile in front of the part's pieces, every other form (the first two groups are the prologue's)
    // as a block at the start.  They have the rest of the step to land: every wave waits for everything at the next barrier.
    const int nfull = max(0, min(n, (B >> 4) - g0));         // groups of the range with 16 spectra
    auto stage3 = [&](int t, const auto &pt) __attribute__((always_inline)) {
        if constexpr (QFA_GT_SETPRIO != 0 && (QFA_GT_PRIO_STAGES & 2) != 0) __builtin_amdgcn_s_setprio(QFA_GT_SETPRIO);
        const bool run = RUNP && runok && t + 2 < nfull;
        if (t + 2 < n && !run) {
            // (batch order: the rows follow from t -- no LDS read; indexed form: stage 2 (t) read them out of the buffer)
            const RowIdx ri = IDX ? ri_next : read_rows(t + 2, t & 1, std::false_type{});
            stage_spectra(t + 2, t & 1, ri, std::false_type{});
        }
        if (!RUNP) part_begin(pt);
        const unsigned char *zp = lds + GT::L_Z + (t & 1) * GT::ZP_B + lane * 16;
        // bf16: three MFMAs per column tile ({l | h} x {h | l}, {h | m} x {m | m}, {h | m} x {h | h}: six products, K = 16 spectra x 2
        // piece slots); float16 (F16S3): two ({h | m} x {h | h} and {h | m} x {m | 0}: three products)
        u32x4 bhl[TPW], bmm[TPW], bhh[TPW];
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            if constexpr (GT::F16S3) {
                unsigned h01, m01, h23, m23;
                split2h(betaR[j][0], betaR[j][1], h01, m01);
                split2h(betaR[j][2], betaR[j][3], h23, m23);
                bhh[j] = u32x4{h01, h23, h01, h23}; bmm[j] = u32x4{m01, m23, 0u, 0u}; bhl[j] = bhh[j];
            } else {
                unsigned h01, m01, l01, h23, m23, l23;
                split2(betaR[j][0], betaR[j][1], h01, m01, l01);
                split2(betaR[j][2], betaR[j][3], h23, m23, l23);
                bhl[j] = u32x4{h01, h23, l01, l23}; bmm[j] = u32x4{m01, m23, m01, m23}; bhh[j] = u32x4{h01, h23, h01, h23};
            }
        }
        // The Z operands of column tile a + ZD are requested in front of the MFMAs of tile a, and fences keep it that way (round 5):
        // left to itself hipcc (which schedules this unit for register pressure) asks for tile a + 1 behind the first MFMA of
        // tile a -- 32 cycles in front of its use, an LDS round trip in the open per column tile: 1 670 cycles for 816 of MFMAs.
#ifndef QFA_GT_ZD
#define QFA_GT_ZD 3
#endif
        constexpr int ZD = QFA_GT_ZD < GT::NWT ? QFA_GT_ZD : GT::NWT, ZR = ZD + 1;
        constexpr int NZ = GT::F16S3 ? 1 : 2;                                                   // 16-byte operands per column tile
        u32x4 zop[ZR][2];
        auto rdz = [&](int a) __attribute__((always_inline)) {                                   // (a == NWT: the p operands, always two)
            const unsigned char *q = zp + (a < GT::NWT ? a * GT::ZT_B : GT::Z_B);
            zop[a % ZR][0] = *reinterpret_cast<const u32x4 *>(q);
            if (NZ == 2 || a == GT::NWT) zop[a % ZR][1] = *reinterpret_cast<const u32x4 *>(q + 1024);
        };
#pragma unroll
        for (int a = 0; a < ZD; ++a) rdz(a);
#pragma unroll
        for (int a = 0; a < GT::NWT; ++a) {
#ifndef QFA_GT_ABL
#define QFA_GT_ABL 0       // timing experiments (wrong results): 1 = stage 3 without its LDS reads, 2 = stage 3 without its DMA pieces
#endif
            if (QFA_GT_ABL & 1) { zop[(a + ZD) % ZR][0] = IBh[0][a % GT::NKQ]; zop[(a + ZD) % ZR][1] = IBm[0][a % GT::NKQ]; }
            else if (a + ZD <= GT::NWT) rdz(a + ZD);
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (RUNP) {
#ifndef QFA_GT_RQS
#define QFA_GT_RQS 2       // column tiles between two staging requests
#endif
                // tiles 0, RQS, 2 RQS, 3 RQS: a request each (the third: blue tiles only); behind them the part's pieces (at most 7)
                constexpr int RQS = QFA_GT_RQS, P0 = 3 * RQS + 1;
                static_assert(P0 + 7 <= GT::NWT + (RQS == 1 ? 2 : 0) || RQS == 1, "piece sites");
                if (a < P0) { if (a % RQS == 0 && run && (a / RQS != 2 || blueTile)) stage_req(t & 1, a / RQS); }
                else if (!(QFA_GT_ABL & 2)) {
                    if (a == P0) part_begin(pt);
                    if (RQS == 1) {
                        if ((a - P0) % 2 == 0) piece(pt, (a - P0) / 2);
                        else if (a == GT::NWT - 1) piece(pt, (a - P0 + 1) / 2);
                    } else if (a - P0 < 7) piece(pt, a - P0);
                }
            } else if (a % 2 == 0 && !(QFA_GT_ABL & 2)) piece(pt, a / 2);
            const u32x4 &Z1 = zop[a % ZR][0], &Z2 = zop[a % ZR][NZ - 1];
#pragma unroll
            for (int j = 0; j < TPW; ++j) {
                if constexpr (GT::F16S3) W[j][a] = xdlh(Z1, bhh[j], xdlh(Z1, bmm[j], W[j][a]));
                else W[j][a] = xdl(Z2, bhh[j], xdl(Z2, bmm[j], xdl(Z1, bhl[j], W[j][a])));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        const u32x4 &P1 = zop[GT::NWT % ZR][0], &P2 = zop[GT::NWT % ZR][1];
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            unsigned h01, m01, l01, h23, m23, l23;
            split2(gamR[j][0], gamR[j][1], h01, m01, l01);
            split2(gamR[j][2], gamR[j][3], h23, m23, l23);
            const u32x4 ghl = {h01, h23, l01, l23}, gmm = {m01, m23, m01, m23}, ghh = {h01, h23, h01, h23};
            gacc[j] = xdl(P2, ghh, xdl(P2, gmm, xdl(P1, ghl, gacc[j])));                            // sum_s p_s[b] gamma[s][px]
        }
        advance_run();
        if constexpr (QFA_GT_SETPRIO != 0 && (QFA_GT_PRIO_STAGES & 2) != 0) __builtin_amdgcn_s_setprio(0);
    };

    // ---- the walk: ONE barrier per group, and the two waves of a SIMD (w, w + 4) a stage apart in the same rotation:
    //   waves 0..3, step t: stage 3 (t)  [Z part of t + 1]   stage 1 (t + 1)  [S1 part of t + 2]   stage 2 (t + 1)
    //   waves 4..7, step t: stage 2 (t)  [Z part of t + 1]   stage 3 (t)      [S1 part of t + 2]   stage 1 (t + 1)
    // so that one wave's VALU phase (stage 2) meets the other's MFMAs.  The barrier of step t says: the Z part of group t
    // and the S1 part of group t + 1 (requested during step t - 1) have landed, and nobody reads Z(t - 1) and S1(t) any
    // more -- their slots take Z(t + 1) and S1(t + 2), piece by piece between the MFMA groups / elements of the stages.
    // Round 5: the staging requests of group t + 2 sit in stage 3 (t) of EVERY wave (one per second column tile, between its
    // MFMAs) and the Z pieces of waves 4..7 in stage 1; every wave waits vmcnt(0) in front of the step's barrier -- nothing stays in
    // flight across it (rounds 3-4: the requests were a block in stage 2 and waves 0..3 left theirs in flight by a counted wait).
    // (Measured forms of this loop, profiles/r3_ablation_pass2.txt: a barrier per half-step with waves 4..7 half a step
    // behind: 7 100 cycles per group -- the stages of a wave ALONE take 5 700 and are no slower beside a partner in another
    // phase, every barrier adds the skew between waves; all eight waves in the same phase, one barrier: 7 400 -- two waves of
    // a SIMD in the same phase do slow each other down.)
    // The stages sit in straight-line loops: under conditions inside one loop hipcc kept two copies of the 64 accumulator
    // registers of W (result of the first MFMA of a chain in fresh registers, copied back at the end of the step).
    if (n > 0) {
        stage_spectra(0, 0, read_rows(0, 0, std::true_type{}), std::true_type{});
        if (n > 1) stage_spectra(1, 1, read_rows(1, 1, std::true_type{}), std::true_type{});
        issue_S1(0);
        if (n > 1) issue_S1(1);
        issue_Z(0);
    }
    dma_wait<0>();
    step_barrier();
    const bool lead = wv8 < 4;
    const NoPart none{};
    if (active && n > 0) {
        stage1(0, none);
        if (lead) stage2(0, none);
    }
    // The parts' source pointers and ring slots run along the walk: the Z part of group t + 1 goes to slot (t + 1) & 1, the S1
    // part of group t + 2 to slot t & 1.  (One walk for every kind of wave: instantiated per (blue, duty) -- six loops -- hipcc
    // ran out of registers and spilled the image pieces.)
    auto walk = [&]() __attribute__((always_inline)) {
        // (biased by PBIAS; a part beyond the range has cnt = 0 and its pointer is never used)
        const unsigned char *zsrc = uniform_ptr(PST + (size_t)(g0 + 1) * GT::STATE_B + GT::S1P_B + z_first * 1024 + PBIAS);
        const unsigned char *ssrc = uniform_ptr(PST + (size_t)(g0 + 2) * GT::STATE_B + s1_first * 1024 + PBIAS);
        const unsigned zd0 = wave_uniform(lds_addr(lds + GT::L_Z + z_first * 1024)) + PBIAS;
        const unsigned sd0 = wave_uniform(lds_addr(lds + GT::L_S1 + s1_first * 1024)) + PBIAS;
        unsigned zdst = zd0 + GT::ZP_B, sdst = sd0;                                        // t = 0: Z(1) -> slot 1, S1(2) -> slot 0
        auto next = [&]() __attribute__((always_inline)) {
            zsrc += GT::STATE_B; ssrc += GT::STATE_B;
            zdst = 2 * zd0 + GT::ZP_B - zdst; sdst = 2 * sd0 + GT::S1P_B - sdst;
        };
        if (lead) {
            for (int t = 0; t < n; ++t) {
                dma_wait<0>();       // (round 5: the staging requests sit in stage 3, the first stage of the step -- nothing is left in flight)
                GTS(0)
                step_barrier();
                GTS(1)
                const Part pz{zsrc, zdst, t + 1 < n ? z_req : 0}, ps{ssrc, sdst, t + 2 < n ? s1_req : 0};
                stage3(t, pz);
                GTS(4)
                if (t + 1 < n) {
                    stage1(t + 1, ps);
                    stage2(t + 1, none);
                    GTS(3)
                }
                next();
            }
        } else {
            for (int t = 0; t < n; ++t) {
                dma_wait<0>();
                GTS(0)
                step_barrier();
                GTS(1)
                const Part pz{zsrc, zdst, t + 1 < n ? z_req : 0}, ps{ssrc, sdst, t + 2 < n ? s1_req : 0};
#ifndef QFA_GT_ZS1
#define QFA_GT_ZS1 1       // waves 4..7: the Z part's pieces between the MFMAs of stage 1 instead of between the elements of stage 2
#endif
                // (a piece issued in stage 2 -- VALU work, nothing to hide behind -- cost the issuing wave ~100 cycles: stage 2 of a RED
                // duty wave, one reciprocal per element, took 1 134 cycles with its seven pieces; between MFMAs they are free)
                if (QFA_GT_ZS1) {
                    stage2(t, none);
                    GTS(3)
                    stage3(t, ps);
                    GTS(4)
                    if (t + 1 < n) stage1(t + 1, pz);
                } else {
                    stage2(t, pz);
                    GTS(3)
                    stage3(t, ps);
                    GTS(4)
                    if (t + 1 < n) stage1(t + 1, none);
                }
                next();
            }
        }
    };
    auto finish = [&]() __attribute__((always_inline)) {
#if QFA_GT_STAMPS
    if (blockIdx.x == 100 && (wv8 == 0 || wv8 == 4) && lane == 0) {
        st_[7] = n;
        for (int i = 0; i < 8; ++i) qfa_gt_stamps[(wv8 >> 2) * 16 + i] = st_[i];
    }
#endif
    dma_wait<0>();

    // ---- the end of the walk: F enters, the sums leave
    if (active) {
#pragma unroll
        for (int j = 0; j < TPW; ++j) {
            const float *fr = reinterpret_cast<const float *>(tile[j] + GT::OFF_F) + lo * KP;
            // F16S3: W holds 2^7 sbeta x the sum (Z as Z 2^zk, beta as beta sbeta 2^(7 - zk)); the gamma term is unscaled
            const float winv = GT::F16S3 ? 0.0078125f / sbeta[j] : 1.f;          // (a power of two: the quotient is exact)
            f32x4 acc = GT::F16S3 ? f32x4{0.f, 0.f, 0.f, 0.f} : gacc[j];
            if constexpr (GT::APT == 1) {
#pragma unroll
                for (int a4 = 0; a4 < KP / 4; ++a4) {
                    const float4 f4 = *reinterpret_cast<const float4 *>(fr + 4 * a4);
                    const float fa[4] = {f4.x, f4.y, f4.z, f4.w};
#pragma unroll
                    for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                        for (int r = 0; r < 4; ++r) acc[r] = fmaf(fa[jj], W[j][4 * a4 + jj][r], acc[r]);
                }
            } else {
                // KP = 8: the lane's rows 4 g + r of tile wtile are (a = 2 wtile + (g >> 1), b = 4 (g & 1) + r); the two a of a tile
                // sit in the lanes g and g ^ 2, summed across them below
#pragma unroll
                for (int wtile = 0; wtile < GT::NWT; ++wtile) {
                    const float fa = fr[2 * wtile + (g >> 1)];
#pragma unroll
                    for (int r = 0; r < 4; ++r) acc[r] = fmaf(fa, W[j][wtile][r], acc[r]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] += __shfl_xor(acc[r], 32);
            }
            if constexpr (GT::F16S3) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = fmaf(acc[r], winv, gacc[j][r]);
            }
            if (inb[j] && (GT::APT == 1 || g < 2)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int b = (GT::APT == 1 ? 4 * g : 4 * (g & 1)) + r;
                    if (b < Nh) {
                        float *q = accF + (size_t)px[j] * Nh + b;
                        if (det) *q = acc[r];
                        else atomicAdd(q, acc[r]);
                    }
                }
            }
            // per-pixel sums over the lanes lo + 16 g'
            float vsA = sA[j], vgPsi = 0.5f * gPsi[j], vgOm = 0.5f * gOm[j], vcnt = (float)cnt[j];
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                vsA += __shfl_xor(vsA, o);
                vgPsi += __shfl_xor(vgPsi, o);
                vgOm += __shfl_xor(vgOm, o);
                vcnt += __shfl_xor(vcnt, o);
            }
            if (inb[j]) {
                // quantity g of the pixel: sumA | gPsi | gOmega (blue pixels) | cnt
                const float v = g == 0 ? vsA : (g == 1 ? vgPsi : (g == 2 ? vgOm : vcnt));
                const bool ok = g != 2 || px[j] < Nb;
                float *q = accA + (g == 0 ? px[j] : (g == 1 ? Npix + px[j] : (g == 2 ? 2 * Npix + px[j] : 2 * Npix + Nb + px[j])));
                if (ok) {
                    if (det) *q = v;
                    else atomicAdd(q, v);
                }
            }
        }
    }
    double s_tau0 = d_tau0, s_c0 = d_c0, s_beta = d_beta;
    for (int o = 32; o >= 1; o >>= 1) {
        s_tau0 += __shfl_xor(s_tau0, o);
        s_c0 += __shfl_xor(s_c0, o);
        s_beta += __shfl_xor(s_beta, o);
    }
    if (det) {
        if (lane == 0) {
            double *q = slabS + ((size_t)blockIdx.x * GT::NW + wv8) * 3;
            q[0] = s_tau0; q[1] = s_c0; q[2] = s_beta;
        }
    } else {
        scal64_commit(sc64, s_tau0, s_c0, s_beta, gridDim.x * (unsigned)GT::NW, accS);
    }
    };
    if (!active) {
        for (int t = 0; t < n; ++t) {
            dma_wait<0>();
            step_barrier();
            if (t + 1 < n) issue_Z(t + 1);
            if (t + 2 < n) issue_S1(t + 2);
        }
    } else walk();
    finish();
}

// qfa_step_kernels.h -- the hot kernels of one QFA training / prediction step on CDNA4 (gfx950).
//
//   k_prep_pf      F, Psi, omega -> PF image (pass 1 B operand, f32 form) and tile-major PFT image (pass 2)
//   k_moments      pass 1, f32 form (N_h > 16): C, T, b, b2 + scalar sums of 16 spectra per wave on
//                  v_mfma_f32_16x16x4_f32 (N_h <= 16 runs k_moments_x of qfa_xdl_kernels.h on the bf16 XDL pipe)
//   k_solve        k x k Gauss-Jordan in fp64 on KP lanes per spectrum (DPP / readlane broadcasts)
//   k_reduce_nll   sum NLL / spectrum counts (fp64, fixed order)
//   k_grads        pass 2: u, diag(Sigma^-1), Psi/omega/scalar sums (stage 1 on the f32 MFMA), F-gradient
//                  contraction (stage 3: XDL pipe with split-bf16 static operands at N_h = 16, f32 MFMA otherwise)
//   k_predict_out  cont = F hmean + mu, unc = sqrt(f^T hcov f)
//
// Both passes share one structure: a 256-thread workgroup = 4 waves = 4 x 16 spectra walks a segment of the
// pixel axis in tiles (16 pixels; 32 in k_moments_x); the tile of the parameter image every wave needs is staged
// once per workgroup in LDS (LDS-DMA or register-staged, one or two tiles ahead, ONE barrier per tile) and the
// spectra of the next tile are prefetched into registers before the arithmetic of the current one.  grid.y splits
// the pixel axis into segments so that (#spectra tiles x #segments) fills 256 CUs evenly (tail quantisation,
// small batches).
#pragma once
#include <type_traits>

#include "qfa_common.h"
#include "qfa_gt_layout.h"       // build_state: k_solve writes the operand images of the pixel-resident pass 2 itself

// ------------------------------------------------------------------------------------------------
template <int KP>
__global__ void k_prep_pf(const float *__restrict__ F, const float *__restrict__ Psi,
                          const float *__restrict__ omega, const float4 *__restrict__ ZP, int Npix, int Nb, int Nh,
                          int NpixPad, float *__restrict__ PF, float *__restrict__ PFT) {
    using C = Cfg<KP>;
    const int i = blockIdx.x * blockDim.y + threadIdx.y;   // pixel row
    if (i >= NpixPad) return;
    const bool live = i < Npix;
    float *pft = PFT + (size_t)(i >> 4) * C::TILE_PFT + (i & 15);
    for (int c = threadIdx.x; c < C::NCPL; c += blockDim.x) {
        float v = 0.f;
        int rt = -1;                       // row in the PFT tile
        if (c < C::FW) {
            if (live && c < Nh) v = F[(size_t)i * Nh + c];
            if (c < KP) rt = c;
        } else if (c < C::PF_PSI) {
            const int pidx = c - C::FW;
            if (pidx < C::KK2) {
                int a = 0;
                while (a + 1 < KP && pair_index(a + 1, a + 1, KP) <= pidx) ++a;
                const int b = a + (pidx - pair_index(a, a, KP));
                if (live && b < Nh) v = F[(size_t)i * Nh + a] * F[(size_t)i * Nh + b];
                rt = KP + pidx;
            }
        } else if (c == C::PF_PSI) {
            if (live) v = Psi[i];
            rt = C::PFT_PSI;
        } else if (c == C::PF_PSI + 1) {
            if (i < Nb) v = omega[i];
            rt = C::PFT_PSI + 1;
        }
        PF[(size_t)i * C::NCPL + c] = v;
        if (rt >= 0) pft[rt * 16] = v;
    }
    // the per-pixel factors of the factored-z form, then zero padding rows of the PFT tile
    for (int r = C::PFT_PSI + 2 + threadIdx.x; r < C::NR; r += blockDim.x) {
        float v = 0.f;
        if (ZP && i < Nb && r < C::PFT_PSI + 5) {
            const float4 q = ZP[i];
            v = r == C::PFT_PSI + 2 ? q.x : (r == C::PFT_PSI + 3 ? q.y : q.z);
        }
        pft[r * 16] = v;
    }
    if constexpr (C::XS3 && KP == 32) {
        // F of this pixel as bf16 pieces, A operand of stage 3 (K = a = 32): [piece][g][px][a = 8g + j], 16 bytes per (g, px)
        if (threadIdx.x < 4) {
            const int g = threadIdx.x;
            unsigned *fp = reinterpret_cast<unsigned *>(PFT + (size_t)(i >> 4) * C::TILE_PFT + C::PFT_MAIN) +
                           (g * 16 + (i & 15)) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int a = 8 * g + 2 * q;
                const float v0 = (live && a < Nh) ? F[(size_t)i * Nh + a] : 0.f;
                const float v1 = (live && a + 1 < Nh) ? F[(size_t)i * Nh + a + 1] : 0.f;
                unsigned h, m, l;
                split2(v0, v1, h, m, l);
                fp[q] = h; fp[256 + q] = m; fp[512 + q] = l;
            }
            if constexpr (QFA_S3_F16 != 0) {
                // the same row as two float16 pieces of t F (k_grads_s3): t from the row's largest |f| (the four threads of a pixel are
                // neighbours in a wave), 1 / t in the KiB behind the pieces
                float mx = 0.f, x[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    x[j] = (live && 8 * g + j < Nh) ? F[(size_t)i * Nh + 8 * g + j] : 0.f;
                    mx = fmaxf(mx, fabsf(x[j]));
                }
                mx = fmaxf(mx, __shfl_xor(mx, 1));
                mx = fmaxf(mx, __shfl_xor(mx, 2));
                float it;
                const float t = f16_row_scale(mx, it);
#pragma unroll
                for (int j = 0; j < 8; ++j) x[j] *= t;
                u32x4 h4, m4;
                split8h(x, h4, m4);
#pragma unroll
                for (int q = 0; q < 4; ++q) { fp[C::PFT_F16H + q] = h4[q]; fp[C::PFT_F16M + q] = m4[q]; }
                if (g == 0) reinterpret_cast<float *>(PFT + (size_t)(i >> 4) * C::TILE_PFT + C::PFT_MAIN)[C::PFT_F16IT + (i & 15)] = it;
            }
        }
    } else if constexpr (C::XS3) {
        // F of this pixel as bf16 pieces, A operand of stage 3: [piece][g][px][a = 4g + j], 8 bytes per (g, px)
        if (threadIdx.x < 4) {
            const int g = threadIdx.x;
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (live && 4 * g + j < Nh) ? F[(size_t)i * Nh + 4 * g + j] : 0.f;
            unsigned h0, m0, l0, h1, m1, l1;
            split2(v[0], v[1], h0, m0, l0);
            split2(v[2], v[3], h1, m1, l1);
            unsigned *fp = reinterpret_cast<unsigned *>(PFT + (size_t)(i >> 4) * C::TILE_PFT + C::PFT_MAIN) +
                           g * 32 + (i & 15) * 2;
            fp[0] = h0; fp[1] = h1;
            fp[128] = m0; fp[129] = m1;
            fp[256] = l0; fp[257] = l1;
        }
    }
}

// copy one contiguous tile (NF4 float4) global -> registers -> LDS with all 256 threads.  The first
// three staging registers are named locals of the kernel (an array or a struct captured by the nested
// lambdas ended up in scratch for N = 3); tiles of N_h > 16 continue in a local array.
template <int NF4>
struct TileCopy {
    static constexpr int N = (NF4 + 255) / 256;
    static constexpr int NX = N > 3 ? N - 3 : 1;
    static __device__ __forceinline__ bool in(int i, int idx) { return 256 * i + 255 < NF4 || idx < NF4; }
    static __device__ __forceinline__ float4 ld(const float4 *__restrict__ src, int i, int tid) {
        return src[in(i, tid + 256 * i) ? tid + 256 * i : NF4 - 1];
    }
    static __device__ __forceinline__ void load(const float4 *__restrict__ src, int tid, float4 &v0, float4 &v1,
                                                float4 &v2, float4 (&vx)[NX]) {
        v0 = ld(src, 0, tid);
        if (N > 1) v1 = ld(src, 1, tid);
        if (N > 2) v2 = ld(src, 2, tid);
#pragma unroll
        for (int i = 3; i < N; ++i) vx[i - 3] = ld(src, i, tid);
    }
    static __device__ __forceinline__ void store(float4 *dst, int tid, const float4 &v0, const float4 &v1,
                                                 const float4 &v2, const float4 (&vx)[NX]) {
        if (in(0, tid)) dst[tid] = v0;
        if (N > 1 && in(1, tid + 256)) dst[tid + 256] = v1;
        if (N > 2 && in(2, tid + 512)) dst[tid + 512] = v2;
#pragma unroll
        for (int i = 3; i < N; ++i)
            if (in(i, tid + 256 * i)) dst[tid + 256 * i] = vx[i - 3];
    }
};

// (Round 1's float32-MFMA pass 1, k_moments, left the library in round 4: k_moments_x (qfa_xdl_kernels.h) runs pass 1 on
// the XDL pipe at every N_h -- 1.25 against 2.24 ms at c3, 1.9 against 3.2 ms at c5.)
struct __attribute__((packed, aligned(4))) f4u { float v[4]; };       // 4-byte aligned 16-byte load
struct __attribute__((packed, aligned(1))) u4u { unsigned char v[4]; };

// ------------------------------------------------------------------------------------------------
// k_sum_segments : MOM[0] += MOM[1] + ... + MOM[nseg-1] (fixed order), float4-vectorised.
// ------------------------------------------------------------------------------------------------
static __global__ void k_sum_segments(float4 *__restrict__ mom, const float4 *__restrict__ rest, int nseg, size_t n4) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n4) return;
    float4 a = mom[i];
    for (int g = 1; g < nseg; ++g) {
        const float4 b = rest[(size_t)(g - 1) * n4 + i];
        a.x += b.x; a.y += b.y; a.z += b.z; a.w += b.w;
    }
    mom[i] = a;
}

// ------------------------------------------------------------------------------------------------
// k_solve.  KP lanes per spectrum, lane c holds column c (= row c) of the symmetric k x k
// matrices in registers; in-place Gauss-Jordan inversion in fp64, every step broadcasting the
// pivot column inside the lane group (DPP row broadcast at KP = 16, wavefront shuffles otherwise).  The pivots are the squared Cholesky diagonal, so
// log det C = sum log(pivot) (finite where the reference's float32 det overflows, QFA/utils.py:54).
// ------------------------------------------------------------------------------------------------
#ifndef QFA_SOLVE_OCC4
#define QFA_SOLVE_OCC4 1      // four waves per SIMD at N_h <= 16 (the state-writing instantiation sat at 130 VGPRs: 126 now, no scratch; c3 solve 0.179 -> 0.163 ms)
#endif
// NLLRED (small batches, training): the block that finishes last also does k_reduce_nll's one-block job -- sum NLL, the number
// of spectra with a blue pixel and B into the packed buffer's scalars `scal`, float64, the same fixed order (bit-identical) --
// through the arrival counter ticket[1], which the step's first kernel (k_prep_step) zeroed: one link less in the chain of
// dependent launches of a small-batch step.
template <int KP, bool PREDICT, bool STATE = false, bool NLLRED = false>
__global__ __launch_bounds__(256, (KP <= 16 && QFA_SOLVE_OCC4) ? 4 : 2) void k_solve(const float *__restrict__ MOM, float *__restrict__ SOL,
                                               float *__restrict__ nll_out, float *__restrict__ nblue_out, int B,
                                               int Nh, float *__restrict__ hmean, float *__restrict__ hcov,
                                               unsigned *__restrict__ ticket = nullptr,
                                               unsigned char *__restrict__ PST = nullptr, float *__restrict__ scal = nullptr) {
    static_assert(!NLLRED || (!PREDICT && !STATE), "NLLRED: the training step of a small batch");
    using C = Cfg<KP>;
    constexpr int G = 64 / KP;
    static_assert(!STATE || ((KP == 16 || KP == 8) && !PREDICT), "state images: N_h <= 16, training step");
    // STATE: the block's 4 G spectra are one (KP = 16) or two (KP = 8) groups of the pixel-resident pass 2 (qfa_grads_t.h):
    // their records go to LDS instead of SOL, and the block turns them into the groups' split-bf16 operand images
    // (build_state, qfa_gt_layout.h)
    __shared__ __attribute__((aligned(16))) float s_rows[STATE ? 4 * G * C::NSOL : 1];
    if (ticket && blockIdx.x == 0 && threadIdx.x == 0) {
        *ticket = 0u;                                                    // arrival counter of k_reduce_nll (this step)
        Scal64 *q = reinterpret_cast<Scal64 *>(ticket + 2);              // float64 scalar-gradient sums of pass 2
        q->s[0] = 0.0; q->s[1] = 0.0; q->s[2] = 0.0; q->ticket = 0u;
    }
    const int lane = threadIdx.x & 63;
    const int c = lane % KP;
    const int s = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * G + lane / KP;
    const bool valid = s < B;
    const float *mom = MOM + (size_t)(valid ? s : B - 1) * C::NMOM;

    // KP = 32: a broadcast inside the 32-lane group costs six instructions per float64 (two v_readlane per dword and
    // a select), and the elimination needs KP^2 of them: the kernel was 17 000 instructions per wave.  The pivot column
    // (and b, y, T below) therefore goes through LDS: the owning lane stores it, every lane reads it back as
    // same-address (broadcast) reads, two float64 per ds_read_b128.  One wave owns its LDS area: LDS executes a wave's
    // operations in order, so a wait for the wave's own stores is all the synchronisation there is.
    constexpr bool VIA_LDS = KP == 32;
    __shared__ __attribute__((aligned(16))) double s_col[VIA_LDS ? 4 * G * KP : 1];
    __shared__ __attribute__((aligned(16))) float s_T[VIA_LDS ? 4 * G * KP * KP : 1];
    double *col = s_col + ((threadIdx.x >> 6) * G + lane / KP) * (VIA_LDS ? KP : 0);
    auto wave_lds_sync = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_wave_barrier();
    };

    double Cc[KP];
    static_for<KP>([&](auto R) {
        constexpr int r = decltype(R)::value;
        const int a = r < c ? r : c, b = r < c ? c : r;
        Cc[r] = (double)mom[pair_index(a, b, KP)] + (r == c ? 1.0 : 0.0);
    });
    // log det C = sum log(pivot): the pivots are multiplied up in float64 (>= 1, far from overflow in groups of
    // eight) and ONE logarithm is taken per group -- log() in float64 was a third of this kernel's instructions
    double logdet = 0.0, pprod = 1.0;
    static_for<KP>([&](auto JJ) {
        constexpr int jj = decltype(JJ)::value;
        // pivot column jj lives in lane jj; every element is broadcast right where it is consumed
        if constexpr (VIA_LDS) {
            if (c == jj) static_for<KP>([&](auto R) { col[decltype(R)::value] = Cc[decltype(R)::value]; });
            wave_lds_sync();
        }
        const double piv = VIA_LDS ? col[jj] : group_bcast<KP>(Cc[jj], jj);
        pprod *= piv;
        if ((jj & 7) == 7 || jj == KP - 1) {
            logdet += log(pprod);
            pprod = 1.0;
        }
        const double ip = 1.0 / piv;
        const double rjc = (c == jj) ? ip : Cc[jj] * ip;
        const double keep = (c == jj) ? 0.0 : 1.0;                 // lane jj: new = -cij * ip (its rjc is ip)
        static_for<KP>([&](auto I) {
            constexpr int i = decltype(I)::value;
            if constexpr (i != jj) {
                const double cij = VIA_LDS ? col[i] : group_bcast<KP>(Cc[i], jj);   // A[i][jj] before this step's update
                Cc[i] = fma(-cij, rjc, Cc[i] * keep);
            }
        });
        Cc[jj] = rjc;
        if constexpr (VIA_LDS) wave_lds_sync();     // the reads of this column are done before the next one is stored
        __builtin_amdgcn_sched_barrier(0);      // keep the broadcasts of step jj+1 out of step jj (VGPR pressure)
    });
    // y = C^-1 b  (Cc[r] = Cinv[r][c] = Cinv[c][r])
    const double bc = (double)mom[C::MOM_B + c];
    double y = 0.0;
    if constexpr (VIA_LDS) {
        col[c] = bc;
        wave_lds_sync();
    }
    static_for<KP>([&](auto R) {
        constexpr int r = decltype(R)::value;
        y += Cc[r] * (VIA_LDS ? col[r] : group_bcast<KP>(bc, r));
    });
    double quad = bc * y;
#pragma unroll
    for (int o = KP / 2; o >= 1; o >>= 1) quad += __shfl_xor(quad, o, KP);
    const float *sc = mom + C::MOM_S;
    const double nll = 0.5 * ((double)sc[0] - quad + (double)sc[2] * (double)QFA_LOG2PI + (double)sc[1] + logdet);
    if (valid && c == 0) {
        nll_out[s] = (float)nll;
        if (nblue_out) nblue_out[s] = sc[3];
    }

    float *sol = STATE ? s_rows + ((threadIdx.x >> 6) * G + lane / KP) * C::NSOL : SOL + (size_t)(valid ? s : 0) * C::NSOL;
    if (valid) {
        sol[c] = (float)y;
        static_for<KP>([&](auto R) {
            constexpr int r = decltype(R)::value;
            if (r <= c) sol[C::SOL_CI + pair_index(r, c, KP)] = (float)(r == c ? Cc[r] : 2.0 * Cc[r]);
        });
    }
    if (PREDICT) {
        if (valid && c < Nh) {
            hmean[(size_t)s * Nh + c] = (float)y;
            static_for<KP>([&](auto R) {
                constexpr int r = decltype(R)::value;
                if (r < Nh) hcov[((size_t)s * Nh + c) * Nh + r] = (float)Cc[r];
            });
        }
        return;
    }
    // T column c (= row c); Z row c: Z[c][b] = sum_m Cinv[c][m] T[m][b] (float32 products of the
    // float64-inverted C^-1: Z is stored in float32 anyway); p_c = b2_c - sum_m T[c][m] y_m
    float Tc[KP];
    const float *momT = mom;
    if constexpr (VIA_LDS) asm volatile("" : "+v"(momT));          // (keeps these 32 loads behind the elimination: registers)
    static_for<KP>([&](auto R) {
        constexpr int r = decltype(R)::value;
        const int a = r < c ? r : c, b = r < c ? c : r;
        Tc[r] = momT[C::MOM_T + pair_index(a, b, KP)];
    });
    float Zr[KP];
    static_for<KP>([&](auto Bq) { Zr[decltype(Bq)::value] = 0.f; });
    double pc = (double)mom[C::MOM_B2 + c];
    if constexpr (VIA_LDS) {
        // T[m][b] for every lane: the group's copy of T in LDS ([m][b], lane c stores column c = row c), read back four
        // floats at a time; y the same way through the column buffer
        float *Tl = s_T + ((threadIdx.x >> 6) * G + lane / KP) * KP * KP;
        wave_lds_sync();                                        // (the reads of b above)
        static_for<KP>([&](auto M) { Tl[decltype(M)::value * KP + c] = Tc[decltype(M)::value]; });
        col[c] = y;
        wave_lds_sync();
        static_for<KP>([&](auto M) {                            // p first: T's registers are free during Z
            constexpr int m = decltype(M)::value;
            pc -= (double)Tc[m] * col[m];
        });
        __builtin_amdgcn_sched_barrier(0);
        static_for<KP>([&](auto M) {
            constexpr int m = decltype(M)::value;
            const float cm = (float)Cc[m];
            static_for<KP / 4>([&](auto Bq) {
                constexpr int b = 4 * decltype(Bq)::value;
                const float4 t = *reinterpret_cast<const float4 *>(Tl + m * KP + b);
                Zr[b] = fmaf(cm, t.x, Zr[b]);
                Zr[b + 1] = fmaf(cm, t.y, Zr[b + 1]);
                Zr[b + 2] = fmaf(cm, t.z, Zr[b + 2]);
                Zr[b + 3] = fmaf(cm, t.w, Zr[b + 3]);
            });
            static_for<KP>([&](auto Bq) { pin(Zr[decltype(Bq)::value]); });   // finish row m before row m + 1 is read
            asm volatile("" ::: "memory");
            __builtin_amdgcn_sched_barrier(0);
        });
    } else {
        static_for<KP>([&](auto M) {
            constexpr int m = decltype(M)::value;
            const float cm = (float)Cc[m];
            static_for<KP>([&](auto Bq) {
                constexpr int b = decltype(Bq)::value;
                Zr[b] = fmaf(cm, group_bcast<KP>(Tc[m], b), Zr[b]);
            });
            static_for<KP>([&](auto Bq) { pin(Zr[decltype(Bq)::value]); });   // finish row m before row m+1's broadcasts
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    if constexpr (!VIA_LDS) {
        static_for<KP>([&](auto M) {
            constexpr int m = decltype(M)::value;
            pc -= (double)Tc[m] * group_bcast<KP>(y, m);
        });
    }
    if (valid) {
        static_for<KP>([&](auto Bq) {
            constexpr int b = decltype(Bq)::value;
            sol[C::SOL_Z + c * KP + b] = Zr[b];
        });
        sol[C::SOL_P + c] = (float)pc;
    }
    if constexpr (STATE) {
        __syncthreads();
#pragma unroll
        for (int q = 0; q < (4 * G) / 16; ++q) {
            const int grp = (int)blockIdx.x * ((4 * G) / 16) + q;
            if (16 * grp < B)
                build_state<KP>(s_rows + q * 16 * C::NSOL, 16 * grp, B, Nh, PST + (size_t)grp * GTT<KP>::STATE_B, (int)threadIdx.x);
        }
    }
    if constexpr (NLLRED) {
        __shared__ double sh[2][4];
        __shared__ bool last;
        __threadfence();                                              // release: this block's nll / nblue before the ticket
        __syncthreads();
        if (threadIdx.x == 0) last = atomicAdd(ticket + 1, 1u) == (unsigned)gridDim.x - 1u;
        __syncthreads();
        if (!last) return;
        __threadfence();                                              // acquire: the other blocks' stores
        double a = 0.0, nb = 0.0;
        for (int q = threadIdx.x; q < B; q += 256) {                  // (k_reduce_nll's order for one block)
            a += (double)__hip_atomic_load(nll_out + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            nb += __hip_atomic_load(nblue_out + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > 0.f ? 1.0 : 0.0;
        }
        for (int o = 32; o >= 1; o >>= 1) {
            a += __shfl_xor(a, o);
            nb += __shfl_xor(nb, o);
        }
        const int w = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { sh[0][w] = a; sh[1][w] = nb; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double ta = 0.0, tb = 0.0;
            for (int i = 0; i < 4; ++i) { ta += sh[0][i]; tb += sh[1][i]; }
            scal[3] += (float)tb;
            scal[4] += (float)ta;
            scal[5] += (float)B;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_reduce_nll : accum scalars += {#spectra with an unmasked blue pixel, sum NLL, B}.  NRED blocks sum one contiguous
// share of the spectra each (float64, fixed order inside a block); the block that arrives last (ticket counter, zeroed by
// k_solve) adds the NRED partial sums in block order: deterministic, and 100 000 spectra no longer pass through one
// block's 1 024 serial load-add chains (39 us).
// ------------------------------------------------------------------------------------------------
constexpr int NRED = 32;
static __global__ __launch_bounds__(256) void k_reduce_nll(const float *__restrict__ nll, const float *__restrict__ nblue,
                                                           int B, float *__restrict__ scal, double *__restrict__ part,
                                                           unsigned *__restrict__ ticket) {
    __shared__ double sh[2][4];
    __shared__ bool last;
    const int per = (B + (int)gridDim.x - 1) / (int)gridDim.x, s0 = blockIdx.x * per, s1 = min(B, s0 + per);
    double a = 0.0, nb = 0.0;
    for (int s = s0 + threadIdx.x; s < s1; s += 256) {
        a += (double)nll[s];
        nb += nblue[s] > 0.f ? 1.0 : 0.0;
    }
    for (int o = 32; o >= 1; o >>= 1) {
        a += __shfl_xor(a, o);
        nb += __shfl_xor(nb, o);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { sh[0][w] = a; sh[1][w] = nb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, tb = 0.0;
        for (int i = 0; i < 4; ++i) { ta += sh[0][i]; tb += sh[1][i]; }
        part[blockIdx.x] = ta;
        part[NRED + blockIdx.x] = tb;
        __threadfence();                                          // release: the partial sums before the ticket
        last = atomicAdd(ticket, 1u) == (unsigned)gridDim.x - 1u;
        if (last) {
            __threadfence();                                      // acquire: the other blocks' partial sums
            ta = 0.0; tb = 0.0;
            for (int i = 0; i < (int)gridDim.x; ++i) {
                ta += __hip_atomic_load(part + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                tb += __hip_atomic_load(part + NRED + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            scal[3] += (float)tb;
            scal[4] += (float)ta;
            scal[5] += (float)B;
        }
    }
}

#if QFA_ABL == 7
__device__ unsigned long long qfa_dbg_stamps[64];
#endif
// ------------------------------------------------------------------------------------------------
// k_grads (pass 2).  Lane (lo = lane&15, g = lane>>4) owns pixel 16*tile + lo of spectra
// s0 + 4g + r (r = 0..3):
//   stage 1  [f^T y | f^T C^-1 f] (16 spectra x 16 px) = [y | Cinv'] (registers, A operand) x PFT tile
//            (LDS, B operand) -> C/D layout col = px, row = 4g + r: the lane's own four elements
//   stage 2  u, diag(Sigma^-1), dG and the Psi / omega / tau0 / c0 / beta sums     (QFA/model.py:136-144)
//   stage 3  accF[px][b] += sum_{s,a} (wD A^2)_{s,px} f_{px,a} Z_s[a][b] + sum_s (A u)_{s,px} p_s[b]
//            K = (spectrum, a); the A operand is lane-local, the B operand Z_{s0+4g+r}[a][b = lo] stays in
//            registers for the whole pixel loop.
// Every wave parks its tile result in its own LDS slot; after the tile barrier one wave sums the
// slots in fixed order and adds the workgroup's tile to the packed accumulation buffer with 5
// full-width float atomics (4 x 256 B contiguous rows of accF + one row of per-pixel sums).
// gF = f * sumA - accF is formed in k_finalize (QFA/model.py:137 in low-rank form, App. A step 7).
// Blue and red tiles run in two specialised loops (the red one has no transcendental work).
// ------------------------------------------------------------------------------------------------
struct SpecRegs2 {
    float d[4], sg[4], z[4];
    unsigned m[4];
};

#ifndef QFA_G8_OCC
#define QFA_G8_OCC 2      // workgroups per CU the N_h <= 8 instantiation is compiled for
#endif
struct PixPar {                  // per-pixel parameters of the lane's pixel: Psi, omega and (factored-z form) ti, pwi, l2i
    float Psi, om, ti, pwi, l2i;
};
template <int KP, bool HASA, bool ZF>
__global__ __launch_bounds__(256, KP > 16 ? 1 : (KP == 8 ? QFA_G8_OCC : 2)) void k_grads(qfa_params_t p, qfa_batch_t bt, qfa_tau_t tau, int B,
                                                              int Npix, int Nb, int Nh, int ntiles, WorkPlan wp,
                                                              int bhalf, const float *__restrict__ PFT,
                                                              const float *__restrict__ SOL,
                                                              float *__restrict__ accum, float *__restrict__ slab,
                                                              double *__restrict__ slabS, int slab_stride,
                                                              Scal64 *__restrict__ sc64, const float4 *__restrict__ ZS,
                                                              float *__restrict__ BG = nullptr,
                                                              float *__restrict__ GG = nullptr, int bg_stride = 0) {
    // BG != NULL (KP = 32): beta = wD A^2 and gamma = A u of every (spectrum, pixel) are also stored, [Bpad][bg_stride]
    // each, for k_grads_s3, which produces the second 16 columns of the F gradient from them: stages 1 and 2 run once.
    // slab != NULL: deterministic mode -- the block's tile partials go to row blk of the slab (plain stores), its
    // scalar sums to slabS[item][wave][3]; k_reduce_slab adds the rows to accum in block order.
    // bhalf: which 16 columns of the F gradient this launch produces (N_h > 16 runs the kernel once per
    // half; the per-pixel and scalar sums are added by the bhalf == 0 launch only)
    using C = Cfg<KP>;
    constexpr int KF = KP / 4, KQ = C::KK2 / 4;
    constexpr int NF4 = C::TILE_PFT / 4;
    constexpr int NPART = 512;      // per wave and tile: aG [px][16] (256) + 4 per-pixel sums x 64 lanes
    constexpr bool XS3 = C::XS3;
    // ring of 3 parameter tiles; with stage 3 on the XDL pipe only the F pieces of a tile are read in its own step
    // (the float32 part is consumed one step earlier by stage 1), so the float32 part needs two slots only
    // KP = 32: stage 3 on the XDL pipe as well, but the float32 part of the tile keeps travelling through the staging
    // registers into a ring of 3 (XDMA = false) and every lane reads its two 16-byte F pieces of the tile straight
    // from the image in global memory (L2) at the start of the step.  (Through LDS -- by LDS-DMA into a ring, or
    // staged with the rest of the tile -- they arrived wrong at this size; not understood, the parity tests caught it.)
    constexpr bool XDMA = XS3 && KP == 16;
    constexpr int NM4 = C::PFT_MAIN / 4, NP4 = XDMA ? C::PFT_FP / 4 : 1, RING_M = XDMA ? 2 : 3;
    constexpr int NCH_MAIN = C::PFT_MAIN / 256, NCH = C::TILE_PFT / 256;   // 1-KiB LDS-DMA pieces (XDL form)
    __shared__ float4 lds4[RING_M][XS3 ? NM4 : NF4];
    __shared__ float4 ldsfp[XDMA ? 3 : 1][NP4];                  // F as bf16 pieces (stage-3 A operand, KP = 16)
    __shared__ unsigned ldszl[XDMA ? 4 : 1][XDMA ? 16 * 64 * 2 : 1];   // third bf16 piece of Z (KP = 16), per wave [spectrum][lane][2]
    __shared__ float ldspart[2][4][NPART];
#if QFA_ABL == 6                                                  // occupancy experiment: one workgroup per CU
    __shared__ float ldspad[22000];
    if (B < 0) ldspad[threadIdx.x] = 1.f;
    if (B < -1) accum[0] = ldspad[threadIdx.x + 1];
#endif
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wv = wave_uniform(tid >> 6);
    int blk, seg, t0, t1;
    plan_item(wp, blockIdx.x, ntiles, blk, seg, t0, t1);
    const int s0 = (blk * 4 + wv) * 16;
    const bool active = s0 < B;
    const int nbt = (Nb + 15) >> 4;
    for (int i = tid; i < 2 * 4 * NPART; i += 256) (&ldspart[0][0][0])[i] = 0.f;   // inactive waves' slots stay 0
    const DevConsts k = load_consts(p, tau);
    const int lo = lane & 15, g = lane >> 4;

    const bool det = slab != nullptr;
    float *accF = det ? slab + (size_t)blk * (size_t)slab_stride : accum;
    float *accA = accF + (size_t)Npix * Nh;
    float *accS = accum + (size_t)Npix * Nh + 3 * (size_t)Npix + Nb;
    auto add_to = [&](float *q, float v) {
        if (det) *q = v;                     // every (block, tile) element is written exactly once
        else atomicAdd(q, v);
    };

    // A operands of stage 1: spectrum s0+lo, k = 4t + g
    float yA[KF], qA[KQ];
    {
        const bool v = active && (s0 + lo) < B;
        const float *sol = SOL + (size_t)(v ? s0 + lo : 0) * C::NSOL;
#pragma unroll
        for (int t = 0; t < KF; ++t) yA[t] = v ? sol[4 * t + g] : 0.f;
#pragma unroll
        for (int t = 0; t < KQ; ++t) qA[t] = v ? sol[C::SOL_CI + 4 * t + g] : 0.f;
    }
    // B operands of stage 3, column b = lo.
    //   f32 MFMA form (N_h > 16): Z of the lane's own spectra s0+4g+r, K = (spectrum, a).
    //   XDL form: G_s = F_tile Z_s per spectrum (both operands static), K = a: the lane holds Z_s[a = 4g+j][b] of
    //   ALL 16 spectra as bf16 pieces h, m (registers) and l (LDS); beta is applied to G_s on the VALU.
    constexpr int NZR = XS3 ? 1 : 4, NZA = XS3 ? 1 : KP, NZS = XS3 ? 16 : 1;
    float Zr[NZR][NZA], pr[4];
    // (KP = 32, THIS kernel's own stage 3 -- the build-time fallback form QFA_P2_S12=0, not k_grads_s3, which issues six products
    // from round 4 on: K = a = 32 per MFMA, 8 values per lane and piece, the two leading pieces only -- four products, <= 2^-17
    // each; the third piece would take 16 KB of LDS per wave)
    using ZV = std::conditional_t<KP == 32, u32x4, u32x2>;
    constexpr int NZJ = KP == 32 ? 8 : 4;              // values of Z per lane and spectrum
    ZV Zh[NZS], Zm[NZS];
    u32x2 ph = {0u, 0u}, pm = {0u, 0u}, pl = {0u, 0u};
    bool sv[4];
    unsigned rowi[4];                             // rows of the lane's four spectra in the batch arrays (qfa_common.h, batch_row); the
    int offB[4];                                  // 64-bit element offsets are formed at the loads (this kernel is at the register limit)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int srel = 4 * g + r;
        sv[r] = active && (s0 + srel) < B;
        const int col = 16 * bhalf + lo;
        const bool v = sv[r] && col < KP;
        const float *sol = SOL + (size_t)(v ? s0 + srel : 0) * C::NSOL;
        if constexpr (!XS3) {
#pragma unroll
            for (int a = 0; a < KP; ++a) Zr[r][a] = v ? sol[C::SOL_Z + a * KP + col] : 0.f;
        }
        pr[r] = v ? sol[C::SOL_P + col] : 0.f;
        const int sc = active ? min(srel, B - 1 - s0) : 0;
        rowi[r] = (unsigned)batch_row(bt, (active ? s0 : 0) + sc);
        offB[r] = sc * Nb;                          // (A_blue: batch order)
    }
    if constexpr (XS3) {
        unsigned *zl = ldszl[KP == 16 ? wv : 0];
        // all 64 loads first (the LDS stores below otherwise serialise them: load 4, wait, split, store, ...)
        float zraw[16][NZJ];
        const int zcol = 16 * bhalf + lo;
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const bool v = active && (s0 + s) < B && zcol < KP;
            const float *sol = SOL + (size_t)(v ? s0 + s : 0) * C::NSOL + C::SOL_Z + zcol;
#pragma unroll
            for (int j = 0; j < NZJ; ++j) zraw[s][j] = (v && NZJ * g + j < KP) ? sol[(NZJ * g + j) * KP] : 0.f;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const float *z = zraw[s];
            if constexpr (KP == 32) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    unsigned h, m, l;
                    split2(z[2 * q], z[2 * q + 1], h, m, l);
                    Zh[s][q] = h;
                    Zm[s][q] = m;
                }
            } else {
                unsigned h0, m0, l0, h1, m1, l1;
                split2(z[0], z[1], h0, m0, l0);
                split2(z[2], z[3], h1, m1, l1);
                Zh[s] = u32x2{h0, h1};
                Zm[s] = u32x2{m0, m1};
                zl[(s * 64 + lane) * 2] = l0;
                zl[(s * 64 + lane) * 2 + 1] = l1;
            }
        }
        // p of the spectra 4g+j as the B operand of the gamma term (K = spectrum)
        unsigned h0, m0, l0, h1, m1, l1;
        split2(pr[0], pr[1], h0, m0, l0);
        split2(pr[2], pr[3], h1, m1, l1);
        ph = u32x2{h0, h1}; pm = u32x2{m0, m1}; pl = u32x2{l0, l1};
    }
    const float *abase = bt.A_blue ? bt.A_blue + (size_t)(active ? s0 : 0) * Nb : nullptr;
    ZFac zs[4];                                   // factored-z form: per-spectrum factors of the lane's four spectra
#pragma unroll
    for (int r = 0; r < 4; ++r) zs[r] = zfac_load(ZS, s0 + 4 * g + r, ZF && sv[r]);
    const float4 *PFT4 = reinterpret_cast<const float4 *>(PFT);
    using TC = TileCopy<XS3 ? NM4 : NF4>;       // (XDL stage 3: the F pieces at the end of the tile are not staged)
    float4 tv0, tv1 = {0.f, 0.f, 0.f, 0.f}, tv2 = {0.f, 0.f, 0.f, 0.f}, tvx[TC::NX];
    double s_tau0 = 0.0, s_c0 = 0.0, s_beta = 0.0;   // float32 per tile, float64 across tiles
    int ntile_done = 0;                               // picks the flushing wave, round robin
#if QFA_ABL == 7      // diagnostic build: where does a tile step spend its cycles (never shipped)
    unsigned long long st_t[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};   // [blue 0..6 | red 8..14]
#define QFA_STAMP(var)                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                      \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory");            \
    __builtin_amdgcn_sched_barrier(0);
#else
#define QFA_STAMP(var)
#endif

    auto run = [&](auto blue_tag, int ta, int tb) {
        constexpr bool BLUE = decltype(blue_tag)::value;
        const int n = tb - ta;
        if (n <= 0) return;                                       // block-uniform
        // de-phase the tile order between workgroups so that concurrent flushes hit different rows
                // (within a window of 64 tiles: the workgroups running together then share 0.8 MB of the tile image in L2)
        const int rot = (int)(((unsigned)blk * 2654435761u) % (unsigned)min(n, 64));
        auto tile_of = [&](int c) {
            int x = c + rot;
            if (x >= n) x -= n;
            return ta + x;
        };

        auto load_spec = [&](int tg, SpecRegs2 &rg) {
            const unsigned px = (unsigned)min(16 * tg + lo, Npix - 1);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                unsigned ri = rowi[r];
                asm volatile("" : "+v"(ri));                            // keep the product out of the loop-invariant set
                const unsigned long long o = (unsigned long long)ri * (unsigned long long)bt.row_stride + px;
                rg.d[r] = bt.delta[o];
                rg.sg[r] = bt.error[o];
                rg.m[r] = bt.mask[o];
            }
            if (BLUE && !ZF) {
                const unsigned pz = (unsigned)min(16 * tg + lo, Nb - 1);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    unsigned ri = rowi[r];
                    asm volatile("" : "+v"(ri));
                    rg.z[r] = bt.zabs[(unsigned long long)ri * (unsigned)Nb + pz];
                }
            }
        };

        // ---- stage 1 of one tile: [f^T y | f^T C^-1 f] for the lane's four elements, plus Psi/omega
        // (plain form, used once to prime the pipeline)
        auto read_pix = [&](const float *tile, PixPar &pp) {
            pp.Psi = tile[C::PFT_PSI * 16 + lo];
            pp.om = tile[(C::PFT_PSI + 1) * 16 + lo];
            if (BLUE && ZF) {
                pp.ti = tile[(C::PFT_PSI + 2) * 16 + lo];
                pp.pwi = tile[(C::PFT_PSI + 3) * 16 + lo];
                pp.l2i = tile[(C::PFT_PSI + 4) * 16 + lo];
            }
        };
        auto stage1 = [&](const float *tile, f32x4 &afy, f32x4 &aq, PixPar &pp) {
            constexpr int NK1 = KF + KQ;
            const float *tb_ = tile + g * 16 + lo;
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int t = 0; t < NK1; ++t) {
                const int row = t < KF ? 4 * t : KP + 4 * (t - KF);
                const float bv = tb_[row * 16];
                if (t < KF) a0 = mfma4(yA[t], bv, a0);
                else if (t & 1) a1 = mfma4(qA[t - KF], bv, a1);
                else a2 = mfma4(qA[t - KF], bv, a2);
            }
            afy = a0;
            aq = a1 + a2;
            read_pix(tile, pp);
        };

        // ---- region 1 of the software pipeline: stage 2 of tile `tg` (VALU: u, diag(Sigma^-1), dG, the
        // per-pixel and scalar sums) hand-woven with stage 1 of the NEXT tile (MFMA).  The per-element
        // arithmetic is cut into chunks of ~6-10 VALU instructions; after each chunk one or two MFMAs of
        // the next tile are issued (their B operands were read from LDS six slots earlier), and
        // sched_barrier(0) pins that order, so a single wave keeps the matrix pipe and the VALU busy
        // at the same time (an MFMA occupies the pipe for 32 cycles = 6-8 VALU issues).
        auto region1 = [&](int tg, const SpecRegs2 &cur, const f32x4 &afy, const f32x4 &aq, const PixPar &pp,
                           const float *tileN, f32x4 &afyN, f32x4 &aqN, PixPar &ppN,
                           float (&betaR)[4], float (&gamR)[4], float *part) {
            constexpr int NK1 = KF + KQ, AHEAD = 6;
            const float *tbN = tileN + g * 16 + lo;
            f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = {0.f, 0.f, 0.f, 0.f}, a2 = {0.f, 0.f, 0.f, 0.f};
            float bop[8];
            auto rd = [&](int t) {
                if (t < NK1) bop[t & 7] = tbN[(t < KF ? 4 * t : KP + 4 * (t - KF)) * 16];
            };
            auto mf = [&](int t) {
                if (t < KF) a0 = mfma4(yA[t < KF ? t : 0], bop[t & 7], a0);
                else if (t < NK1 && (t & 1)) a1 = mfma4(qA[t < NK1 ? t - KF : 0], bop[t & 7], a1);
                else if (t < NK1) a2 = mfma4(qA[t < NK1 ? t - KF : 0], bop[t & 7], a2);
            };
            constexpr int SF = (NK1 + 39) / 40;      // MFMAs per slot (1 up to N_h = 16, 4 at N_h = 32)
            int slot_no = 0;
            auto slots = [&](int nslot) {            // nslot MFMA slots, then fence
#pragma unroll
                for (int i = 0; i < nslot * SF; ++i) {
                    rd(slot_no + AHEAD);
                    mf(slot_no);
                    ++slot_no;
                }
                __builtin_amdgcn_sched_barrier(0);
            };
#pragma unroll
            for (int t = 0; t < AHEAD; ++t) rd(t);
            read_pix(tileN, ppN);
            __builtin_amdgcn_sched_barrier(0);
            const float Psi = pp.Psi, om = pp.om;

            const int px = 16 * tg + lo;
            const bool inb = px < Npix;
            const bool blue = px < Nb;
            float gPsi = 0.f, gOm = 0.f, sA = 0.f, cnt = 0.f;
            float t_tau0 = 0.f, t_c0 = 0.f, t_beta = 0.f;
            // two elements per chunk (independent dependency chains back to back), two MFMA slots after it
            float dd[4], l2[4], x1[4], x2[4], pw[4], y1[4], y2[4], Av[4], zd[4], A2[4], wD[4], wDA[4], uu[4], dG[4];
            bool wv_[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dd[r] = cur.d[r];
                wv_[r] = inb & sv[r] & (cur.m[r] != 0);
            }
#pragma unroll
            for (int rp = 0; rp < 4; rp += 2) {
                if (BLUE) {
#pragma unroll
                    for (int r = rp; r < rp + 2; ++r) {           // chunk 0
                        if (ZF) {                                 // factored-z form (qfa_common.h, ZFac): no logarithm
                            l2[r] = zs[r].l2 + pp.l2i;
                            pw[r] = zs[r].pw * pp.pwi;
                            pin(l2[r], pw[r]);
                        } else {
                            l2[r] = fast_log2(1.0f + cur.z[r]);
                            x1[r] = k.t_expo * (l2[r] + k.t_lscale);
                            x2[r] = k.beta * l2[r];
                            pin(x1[r], x2[r]);
                        }
                    }
                    slots(2);
#pragma unroll
                    for (int r = rp; r < rp + 2; ++r) {           // chunk 1
                        if (ZF) {
                            y1[r] = fmaf(zs[r].ts, pp.ti, k.offp);
                            y2[r] = k.k1 * pw[r];
                        } else {
                            pw[r] = fast_exp2(x2[r]);
                            const float tauv = k.t_amp * fast_exp2(x1[r]) + k.t_off;       // QFA/utils.py:105-141
                            y1[r] = -tauv * QFA_LOG2E;
                            y2[r] = -k.tau0 * pw[r] * QFA_LOG2E;
                        }
                        pin(y1[r], y2[r]);
                    }
                    slots(2);
#pragma unroll
                    for (int r = rp; r < rp + 2; ++r) {           // chunk 2
                        float Ab = fast_exp2(y1[r]);                                   // QFA/model.py:125
                        if (HASA) Ab = abase[offB[r] + min(px, Nb - 1)];               // custom tau callable
                        const float re = 1.0f - k.c0 - fast_exp2(y2[r]);               // QFA/utils.py:91
                        Av[r] = blue ? Ab : 1.f;
                        zd[r] = blue ? re * re : 0.f;
                        pin(Av[r], zd[r]);
                    }
                    slots(2);
#pragma unroll
                    for (int r = rp; r < rp + 2; ++r) {           // chunk 3
                        A2[r] = Av[r] * Av[r];
                        const float D = A2[r] * Psi + om * zd[r] + cur.sg[r] * cur.sg[r];
                        wD[r] = wv_[r] ? fast_rcp(D) : 0.f;
                        dd[r] = wv_[r] ? dd[r] : 0.f;
                        wDA[r] = wD[r] * Av[r];
                        pin(wDA[r], dd[r]);
                    }
                    slots(4);
#pragma unroll
                    for (int r = rp; r < rp + 2; ++r) {           // chunk 4
                        uu[r] = wD[r] * (dd[r] - Av[r] * afy[r]);                      // (Sigma^-1 delta)_i
                        const float dS = wD[r] - wDA[r] * wDA[r] * aq[r];              // diag(Sigma^-1)_i
                        dG[r] = 0.5f * (dS - uu[r] * uu[r]);                           // QFA/model.py:136,138
                        gPsi += A2[r] * dG[r];                                         // :139
                        gOm += dG[r] * zd[r];                                          // :140
                    }
                    pin(gPsi, gOm);
                    slots(4);
#pragma unroll
                    for (int r = rp; r < rp + 2; ++r) {           // chunk 5
                        const float root = 1.0f - k.tau0 * pw[r] - k.c0;               // :141
                        const float e = dG[r] * (om * zd[r]) * zd[r] * 2.0f * root;
                        t_tau0 -= e * pw[r];                                           // :142
                        t_beta -= e * (k.tau0 * pw[r] * (l2[r] * QFA_LN2));            // :143
                        t_c0 -= e;                                                     // :144
                    }
                    pin(t_tau0, t_beta, t_c0);
                    slots(4);
#pragma unroll
                    for (int r = rp; r < rp + 2; ++r) {           // chunk 6
                        cnt += wv_[r] ? 1.f : 0.f;
                        betaR[r] = wDA[r] * Av[r];
                        sA += betaR[r] * Av[r];
                        gamR[r] = Av[r] * uu[r];
                        pin(gamR[r]);
                    }
                    pin(cnt, sA);
                    slots(2);
                } else {                                                               // red side: A = 1, zd = 0
#pragma unroll
                    for (int r = rp; r < rp + 2; ++r) {
                        const float D = Psi + cur.sg[r] * cur.sg[r];
                        wD[r] = wv_[r] ? fast_rcp(D) : 0.f;
                        dd[r] = wv_[r] ? dd[r] : 0.f;
                        pin(wD[r], dd[r]);
                    }
                    slots(6);
#pragma unroll
                    for (int r = rp; r < rp + 2; ++r) {
                        uu[r] = wD[r] * (dd[r] - afy[r]);
                        const float dS = wD[r] - wD[r] * wD[r] * aq[r];
                        gPsi += 0.5f * (dS - uu[r] * uu[r]);
                    }
                    pin(gPsi);
                    slots(6);
#pragma unroll
                    for (int r = rp; r < rp + 2; ++r) {
                        cnt += wv_[r] ? 1.f : 0.f;
                        betaR[r] = wD[r];
                        sA += wD[r];
                        gamR[r] = uu[r];
                        pin(gamR[r]);
                    }
                    pin(cnt, sA);
                    slots(8);
                }
            }
            static_assert(NK1 <= 40 * SF && AHEAD < 8, "stage 1 has more K-steps than MFMA slots");
            if (BLUE) {
                s_tau0 += (double)t_tau0;
                s_c0 += (double)t_c0;
                s_beta += (double)t_beta;
            }
            afyN = a0;
            aqN = a1 + a2;
            part[256 + lane] = sA;
            part[320 + lane] = gPsi;
            part[384 + lane] = gOm;
            part[448 + lane] = cnt;
        };

        // ---- stage 3 of one tile: the F-gradient contraction, f_{px,a} re-read from the tile image
        auto stage3 = [&](const float *tile, const float (&betaR)[4], const float (&gamR)[4], float *part) {
          if constexpr (!XS3) {
            float f[KP];
#pragma unroll
            for (int a = 0; a < KP; ++a) f[a] = tile[a * 16 + lo];
            f32x4 aG = {0.f, 0.f, 0.f, 0.f}, aG2 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
#pragma unroll
                for (int a = 0; a < KP; a += 2) {
                    aG = mfma4(betaR[r] * f[a], Zr[r][a], aG);
                    aG2 = mfma4(betaR[r] * f[a + 1], Zr[r][a + 1], aG2);
                }
                if (r & 1) aG = mfma4(gamR[r], pr[r], aG);
                else aG2 = mfma4(gamR[r], pr[r], aG2);
            }
            aG += aG2;
            // aG: col = b = lo, row = 4g + rr -> pixel 4g + rr
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) part[(4 * g + rr) * 16 + lo] = aG[rr];
          }
        };

        // ---- stage 3 on the XDL pipe: per spectrum G_s = F_tile Z_s (16 px x 16 b, K = a, six bf16 MFMAs over
        // static operands), then accF[px][b] += beta_{s,px} G_s[px][b] on the VALU; beta reaches the lane that
        // holds rows px = 4g..4g+3 of G through the wave's LDS slot.  The gamma term sum_s gamma_{s,px} p_s[b] is
        // one more product with K = spectrum (gamma split in the loop: 4 values per lane).
        auto stage3x = [&](const float4 *fpt, const ZV &Fgh, const ZV &Fgm, const float (&betaR)[4], const float (&gamR)[4],
                           float *part) {
            if constexpr (XS3) {
#pragma unroll
                for (int r = 0; r < 4; ++r) part[(4 * g + r) * 16 + lo] = betaR[r];
                const ZV *fp = reinterpret_cast<const ZV *>(fpt) + lane;
                const ZV Fh = XDMA ? fp[0] : Fgh, Fm = XDMA ? fp[64] : Fgm;
                unsigned h0, m0, l0, h1, m1, l1;
                split2(gamR[0], gamR[1], h0, m0, l0);
                split2(gamR[2], gamR[3], h1, m1, l1);
                const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
                f32x4 acc = xdl16_6(u32x2{h0, h1}, u32x2{m0, m1}, u32x2{l0, l1}, ph, pm, pl, zero);
                const u32x2 *zl = reinterpret_cast<const u32x2 *>(ldszl[KP == 16 ? wv : 0]) + lane;
                const float4 *brow = reinterpret_cast<const float4 *>(part) + g;
#pragma unroll
                for (int s = 0; s < 16; ++s) {
                    f32x4 G;
                    if constexpr (KP == 32) {
                        G = xdl(Fm, Zm[s], zero);
                        G = xdl(Fm, Zh[s], G);
                        G = xdl(Fh, Zm[s], G);
                        G = xdl(Fh, Zh[s], G);
                    } else {
                        G = xdl16_6(Fh, Fm, fp[128], Zh[s], Zm[s], zl[s * 64], zero);
                    }
                    const float4 bq = brow[s * 4];
                    acc[0] = fmaf(bq.x, G[0], acc[0]);
                    acc[1] = fmaf(bq.y, G[1], acc[1]);
                    acc[2] = fmaf(bq.z, G[2], acc[2]);
                    acc[3] = fmaf(bq.w, G[3], acc[3]);
                }
#pragma unroll
                for (int rr = 0; rr < 4; ++rr) part[(4 * g + rr) * 16 + lo] = acc[rr];
            }
        };

        // tile tg leaves the workgroup: every wave sums one quarter of the accF tile over the four
        // waves' slots (fixed order; slots of inactive waves hold zeros) and adds it to the packed
        // buffer with ONE full-width float atomic (256 contiguous bytes at N_h = 16); the wave whose
        // turn it is also adds the row of per-pixel sums [sumA | gPsi | gOmega | cnt] (branch-free).
        auto flush = [&](int tg, const float (*pp)[NPART], bool extra) {
            const int base = 16 * tg;
            {
                const int idx = lane + 64 * wv;
                const float v = (pp[0][idx] + pp[1][idx]) + (pp[2][idx] + pp[3][idx]);
                const int px = base + (idx >> 4), b = 16 * bhalf + (idx & 15);
                if ((b < Nh) & (px < Npix)) add_to(accF + (size_t)px * Nh + b, v);
            }
            if (extra & (bhalf == 0)) {
                const int which = lane >> 4, px = base + (lane & 15);
                float v = 0.f;
#pragma unroll
                for (int w = 0; w < 4; ++w) {
                    const float *q = pp[w] + 256 + which * 64 + (lane & 15);
                    v += (q[0] + q[16]) + (q[32] + q[48]);
                }
                // accA = sumA (Npix) | gPsi (Npix) | gOmega (Nb) | cnt (Npix), contiguous
                const int off = which * Npix - (which == 3 ? Npix - Nb : 0) + px;
                const bool ok = (px < Npix) & ((which != 2) | (px < Nb));
                if (ok) add_to(accA + off, v);
            }
        };
        auto tilebuf = [&](int c) { return reinterpret_cast<const float *>(lds4[c % RING_M]); };
        // parameter tile c -> LDS.  XDL form: LDS-DMA, wave w moves the 1-KiB pieces w, w+4, ... (no staging
        // registers; the tile barrier retires them).  f32 form: through the staging registers (get / put).
        auto get_tile = [&](int c) {
            if constexpr (XDMA) {
                const unsigned char *src = reinterpret_cast<const unsigned char *>(PFT4 + (size_t)tile_of(c) * NF4) + lane * 16;
#pragma unroll
                for (int i = 0; i < (NCH + 3) / 4; ++i) {
                    const int ch = wv + 4 * i;
                    if (ch < NCH_MAIN) glds16(src + ch * 1024, reinterpret_cast<unsigned char *>(lds4[c % RING_M]) + ch * 1024);
                    else if (ch < NCH) glds16(src + ch * 1024, reinterpret_cast<unsigned char *>(ldsfp[c % 3]) + (ch - NCH_MAIN) * 1024);
                }
            } else {
                TC::load(PFT4 + (size_t)tile_of(c) * NF4, tid, tv0, tv1, tv2, tvx);
            }
        };
        auto put_tile = [&](int c) {
            if constexpr (!XDMA) TC::store(lds4[c % RING_M], tid, tv0, tv1, tv2, tvx);
        };

        // Software pipeline over tiles (one barrier per tile, ring of 3 parameter tiles):
        //   step c:  region 1 = stage 1 of tile c+1 (38 MFMAs, independent) beside stage 2 of tile c (VALU),
        //            region 2 = stage 3 of tile c (68 MFMAs);
        // so the matrix pipe has work from a neighbouring tile while the VALU does the per-pixel math.
        f32x4 afy, aq;
        PixPar pxp{0.f, 0.f, 0.f, 0.f, 0.f};
        auto step = [&](int c, const SpecRegs2 &cur, SpecRegs2 &nxt) {
            const bool more = c + 1 < n;
            const int tg = tile_of(c);
            const int pbuf = c & 1;
#if QFA_ABL == 7
            unsigned long long q0 = 0, q1 = 0, q2 = 0, q3 = 0, q4 = 0, q5 = 0, q6 = 0, q7 = 0;
#endif
            QFA_STAMP(q0)
            f32x4 afyN = {0.f, 0.f, 0.f, 0.f}, aqN = {0.f, 0.f, 0.f, 0.f};
            PixPar pxpN{0.f, 0.f, 0.f, 0.f, 0.f};
            float betaR[4] = {0.f, 0.f, 0.f, 0.f}, gamR[4] = {0.f, 0.f, 0.f, 0.f};
            ZV Fgh = {}, Fgm = {};
            if constexpr (XS3 && !XDMA) {      // KP = 32: the lane's F pieces of this tile straight from the tile image (L2)
                if (active) {
                    const ZV *fg = reinterpret_cast<const ZV *>(PFT + (size_t)tg * C::TILE_PFT + C::PFT_MAIN) + lane;
                    Fgh = fg[0];
                    Fgm = fg[64];
                }
            }
            if (active) {
                const int cn1 = more ? c + 1 : c;                 // last tile: harmless recomputation
#if QFA_ABL != 2
                load_spec(tile_of(cn1), nxt);
#else
                nxt = cur;
#endif
                QFA_STAMP(q1)
                region1(tg, cur, afy, aq, pxp, tilebuf(cn1), afyN, aqN, pxpN, betaR, gamR,
                        ldspart[pbuf][wv]);
                __builtin_amdgcn_sched_barrier(0);
                QFA_STAMP(q2)
            }
            // parameter tile c+2: issued here so that its latency runs under stage 3
            if (c + 2 < n) get_tile(c + 2);
            if (active) {
#if QFA_ABL != 5
                if (BG) {                                  // (uniform; 64-byte segments: 16 pixels of one spectrum)
                    const size_t o = (size_t)(s0 + 4 * g) * bg_stride + 16 * tg + lo;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        BG[o + (size_t)r * bg_stride] = betaR[r];
                        GG[o + (size_t)r * bg_stride] = gamR[r];
                    }
                }
                if constexpr (XS3) stage3x(ldsfp[XDMA ? c % 3 : 0], Fgh, Fgm, betaR, gamR, ldspart[pbuf][wv]);
                else stage3(tilebuf(c), betaR, gamR, ldspart[pbuf][wv]);
#else
                asm volatile("" ::"v"(betaR[0]), "v"(betaR[1]), "v"(betaR[2]), "v"(betaR[3]), "v"(gamR[0]), "v"(gamR[1]),
                             "v"(gamR[2]), "v"(gamR[3]));
#endif
                afy = afyN; aq = aqN; pxp = pxpN;
                QFA_STAMP(q3)
#if QFA_ABL == 7
                st_t[(BLUE ? 0 : 8) + 0] += q1 - q0; st_t[(BLUE ? 0 : 8) + 1] += q2 - q1; st_t[(BLUE ? 0 : 8) + 2] += q3 - q2;
#endif
            }
            QFA_STAMP(q4)
            if (c + 2 < n) put_tile(c + 2);
            QFA_STAMP(q5)
            __syncthreads();
            QFA_STAMP(q6)
            // ldspart[pbuf] is rewritten two tiles later, i.e. after the next barrier
#if QFA_ABL != 4 && !defined(QFA_NOFLUSH)
            flush(tg, ldspart[pbuf], wv == (ntile_done & 3));
#endif
            QFA_STAMP(q7)
#if QFA_ABL == 7
            st_t[(BLUE ? 0 : 8) + 3] += q5 - q4; st_t[(BLUE ? 0 : 8) + 4] += q6 - q5; st_t[(BLUE ? 0 : 8) + 5] += q7 - q6;
            st_t[(BLUE ? 0 : 8) + 6] += 1;
#endif
            ++ntile_done;
        };

        SpecRegs2 ra, rb;
        get_tile(0);
        put_tile(0);
        if (n > 1) {
            get_tile(1);
            put_tile(1);
        }
        if (active) load_spec(tile_of(0), ra);
        __syncthreads();
        if (active) stage1(tilebuf(0), afy, aq, pxp);
        for (int c = 0; c < n; c += 2) {
            step(c, ra, rb);
            if (c + 1 < n) step(c + 1, rb, ra);
        }
        __syncthreads();      // the last flush reads ldspart; the next range's first tile rewrites it
    };
    run(std::true_type{}, t0, min(t1, nbt));
    run(std::false_type{}, max(t0, nbt), t1);

#if QFA_ABL == 7
    if (blk == 300 && lane == 0 && wv == 0 && seg < 4) {
        for (int i = 0; i < 16; ++i) qfa_dbg_stamps[seg * 16 + i] = st_t[i];
    }
#endif
    if (!active) {
        if (det && lane == 0 && bhalf == 0) {              // the reducer reads every (item, wave) record
            double *q = slabS + ((size_t)blockIdx.x * 4 + wv) * 3;
            q[0] = 0.0; q[1] = 0.0; q[2] = 0.0;
        } else if (!det && bhalf == 0) {
            scal64_commit(sc64, 0.0, 0.0, 0.0, gridDim.x * 4u, accS);
        }
        return;
    }
    for (int o = 32; o >= 1; o >>= 1) {
        s_tau0 += __shfl_xor(s_tau0, o);
        s_c0 += __shfl_xor(s_c0, o);
        s_beta += __shfl_xor(s_beta, o);
    }
    if (lane == 0 && bhalf == 0) {
        if (det) {
            double *q = slabS + ((size_t)blockIdx.x * 4 + wv) * 3;
            q[0] = s_tau0; q[1] = s_c0; q[2] = s_beta;
        } else {
            scal64_commit(sc64, s_tau0, s_c0, s_beta, gridDim.x * 4u, accS);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Deterministic mode, the fixed-order reduction of the slab (rows of `stride` floats, one per block of 64 spectra):
//   k_reduce_slab_rows : part[chunk][j] = sum of rows [32 chunk, 32 chunk + 32) of column j, rows in order, float64;
//   k_reduce_slab_fin  : accum[j] += sum of the chunks in order; the three scalar gradients from slabS[item][wave]
//                        in item order.
// The summation order depends on the batch size only -- never on timing.
// ------------------------------------------------------------------------------------------------
#ifndef QFA_DET_CHUNK_ROWS
#define QFA_DET_CHUNK_ROWS 32
#endif
static __global__ __launch_bounds__(256) void k_reduce_slab_rows(const float *__restrict__ slab, int nblk, size_t NF,
                                                                 size_t stride, double *__restrict__ part) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= NF) return;
    const int r0 = blockIdx.y * QFA_DET_CHUNK_ROWS, r1 = min(nblk, r0 + QFA_DET_CHUNK_ROWS);
    const float *q = slab + (size_t)r0 * stride + j;
    double a = 0.0;
    int r = r0;
    for (; r + 8 <= r1; r += 8, q += 8 * stride) {               // eight loads in flight, added in row order
        const float v0 = q[0], v1 = q[stride], v2 = q[2 * stride], v3 = q[3 * stride], v4 = q[4 * stride],
                    v5 = q[5 * stride], v6 = q[6 * stride], v7 = q[7 * stride];
        a += (double)v0; a += (double)v1; a += (double)v2; a += (double)v3;
        a += (double)v4; a += (double)v5; a += (double)v6; a += (double)v7;
    }
    for (; r < r1; ++r, q += stride) a += (double)q[0];
    part[(size_t)blockIdx.y * NF + j] = a;
}
static __global__ __launch_bounds__(256) void k_reduce_slab_fin(const double *__restrict__ part,
                                                                const double *__restrict__ slabS, int nch,
                                                                int nitemwaves, size_t NF, float *__restrict__ accum) {
    const size_t j = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (j < NF) {
        double a = 0.0;
        for (int c = 0; c < nch; ++c) a += part[(size_t)c * NF + j];
        accum[j] += (float)a;
    }
    // the three scalar gradients: thread t sums its contiguous share of the records in order, thread k < 3 then the 256
    // shares in order
    __shared__ double sh[3][256];
    if (blockIdx.x == 0) {
        const int per = (nitemwaves + 255) / 256, q0 = threadIdx.x * per, q1 = min(nitemwaves, q0 + per);
        double a0 = 0.0, a1 = 0.0, a2 = 0.0;
        for (int q = q0; q < q1; ++q) {
            a0 += slabS[(size_t)q * 3 + 0];
            a1 += slabS[(size_t)q * 3 + 1];
            a2 += slabS[(size_t)q * 3 + 2];
        }
        sh[0][threadIdx.x] = a0; sh[1][threadIdx.x] = a1; sh[2][threadIdx.x] = a2;
        __syncthreads();
        if (threadIdx.x < 3) {
            double a = 0.0;
            for (int t = 0; t < 256; ++t) a += sh[threadIdx.x][t];
            accum[NF + threadIdx.x] += (float)a;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_predict_out : cont = F hmean + mu on ALL pixels, unc = sqrt(diag(F hcov F^T))  (QFA/model.py:180)
// -- stage 1 of k_grads with [hmean | hcov'] as the A operand; bound by the (B, Npix) output writes.
// ------------------------------------------------------------------------------------------------
template <int KP>
__global__ __launch_bounds__(256) void k_predict_out(const float *__restrict__ mu, int B, int Npix, int ntiles,
                                                     const float *__restrict__ PFT, const float *__restrict__ SOL,
                                                     float *__restrict__ cont, float *__restrict__ unc) {
    using C = Cfg<KP>;
    constexpr int KF = KP / 4, KQ = C::KK2 / 4;
    const int lane = threadIdx.x & 63;
    const int s0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16;
    if (s0 >= B) return;
    const int lo = lane & 15, g = lane >> 4;
    float yA[KF], qA[KQ];
    {
        const bool v = (s0 + lo) < B;
        const float *sol = SOL + (size_t)(v ? s0 + lo : 0) * C::NSOL;
#pragma unroll
        for (int t = 0; t < KF; ++t) yA[t] = v ? sol[4 * t + g] : 0.f;
#pragma unroll
        for (int t = 0; t < KQ; ++t) qA[t] = v ? sol[C::SOL_CI + 4 * t + g] : 0.f;
    }
    for (int tg = 0; tg < ntiles; ++tg) {
        const int px = 16 * tg + lo;
        f32x4 afy = {0.f, 0.f, 0.f, 0.f}, aq = {0.f, 0.f, 0.f, 0.f};
        const float *tb = PFT + (size_t)tg * C::TILE_PFT + g * 16 + lo;
#pragma unroll
        for (int t = 0; t < KF; ++t) afy = mfma4(yA[t], tb[(4 * t) * 16], afy);
#pragma unroll
        for (int t = 0; t < KQ; ++t) aq = mfma4(qA[t], tb[(KP + 4 * t) * 16], aq);
        if (px < Npix) {
            const float m = mu[px];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int s = s0 + 4 * g + r;
                if (s < B) {
                    cont[(size_t)s * Npix + px] = afy[r] + m;
                    unc[(size_t)s * Npix + px] = __builtin_amdgcn_sqrtf(aq[r]);
                }
            }
        }
    }
}

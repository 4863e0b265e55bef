// qfa_kernels.hip -- hand-written CDNA4 (gfx950) kernels of the QFA training / prediction step.
//
// Pipeline of one qfa_nll_grad_f32 call (reference QFA/model.py:74-158 for a whole batch):
//   k_prep_pf      F -> PF image [f | f_a f_b] and its transpose                (once per step)
//   k_moments      stream the spectra once: C, T, b, b2 and the scalar sums, 16 spectra per wave,
//                  contractions on v_mfma_f32_16x16x4_f32                                (pass 1)
//   k_solve        per spectrum k x k Gauss-Jordan in fp64 on KP lanes with shuffles: NLL, y,
//                  C^-1, Z = C^-1 T, p = b2 - T y
//   k_reduce_nll   sum NLL / spectrum counts into the packed accumulation buffer
//   k_grads        stream the spectra again: u, diag(Sigma^-1), the Psi/omega/scalar sums and the
//                  F-gradient contraction over (spectrum, a) on MFMA                     (pass 2)
// Prediction (QFA/model.py:160-180) = k_moments (delta = flux - mu A) + k_solve + k_predict_out.
#include "qfa_common.h"

// ------------------------------------------------------------------------------------------------
// k_prep_pf : PF[i][0..KP) = F[i][:] (zero padded), PF[i][FW + pair(a,b)] = f_ia f_ib;  PFT = the
// same numbers column-major ([c][i], c in [0,KP) for F, [KP, KP+KK2) for pairs).  Rows
// i in [Npix, NpixPad) are zero so that MFMA operand loads never need a bounds check.
// ------------------------------------------------------------------------------------------------
template <int KP>
__global__ void k_prep_pf(const float *__restrict__ F, int Npix, int Nh, int NpixPad,
                          float *__restrict__ PF, float *__restrict__ PFT) {
    using C = Cfg<KP>;
    const int i = blockIdx.x * blockDim.y + threadIdx.y;   // pixel
    if (i >= NpixPad) return;
    const bool live = i < Npix;
    for (int c = threadIdx.x; c < C::NCP; c += blockDim.x) {
        float v = 0.f;
        int ct = -1;
        if (c < C::FW) {
            if (live && c < Nh) v = F[(size_t)i * Nh + c];
            if (c < KP) ct = c;
        } else {
            int pidx = c - C::FW;
            if (pidx < C::KK2) {
                // invert pair_index: find a with pair_index(a,a) <= pidx
                int a = 0;
                while (a + 1 < KP && pair_index(a + 1, a + 1, KP) <= pidx) ++a;
                int b = a + (pidx - pair_index(a, a, KP));
                if (live && b < Nh) v = F[(size_t)i * Nh + a] * F[(size_t)i * Nh + b];
                ct = KP + pidx;
            }
        }
        PF[(size_t)i * C::NCP + c] = v;
        if (ct >= 0) PFT[(size_t)ct * NpixPad + i] = v;
    }
}

// ------------------------------------------------------------------------------------------------
// k_moments (pass 1).  One wave = 16 spectra.  Lane (sl = lane&15, j = lane>>4) owns spectrum
// s0+sl at pixels base + 4j + e (e = 0..3) of every 16-pixel group; that is exactly the A-operand
// layout of v_mfma_f32_16x16x4_f32 (A[i = lane&15][k = lane>>4]) for K-step e, with the B operand
// B[k = lane>>4][col = lane&15] = PF[base + 4j + e][col].  No LDS, no cross-lane traffic in the loop.
// ------------------------------------------------------------------------------------------------
template <int KP, bool PREDICT>
__global__ __launch_bounds__(256) void k_moments(qfa_params_t p, qfa_batch_t bt, qfa_tau_t tau,
                                                 const float *__restrict__ mu, int B, int Npix, int Nb,
                                                 const float *__restrict__ PF, float *__restrict__ MOM) {
    using C = Cfg<KP>;
    const int lane = threadIdx.x & 63;
    const int s0 = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 16;
    if (s0 >= B) return;
    const DevConsts k = load_consts(p, tau);
    const int sl = lane & 15, j = lane >> 4;
    const int s = s0 + sl;
    const bool svalid = s < B;
    const size_t rowN = (size_t)(svalid ? s : B - 1) * Npix;
    const size_t rowB = (size_t)(svalid ? s : B - 1) * Nb;

    f32x4 accC[C::NT], accT[C::NT], accb[C::NFT], accb2[C::NFT];
#pragma unroll
    for (int t = 0; t < C::NT; ++t) accC[t] = accT[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < C::NFT; ++t) accb[t] = accb2[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    // scalar sums: float32 inside a 4-pixel group, float64 across groups (a 1000-term float32
    // chain of log D costs ~1e-5 of the NLL; this keeps it at the 1e-7 level for 1 DADD / 4 px)
    double qd = 0.0, ld = 0.0;
    float cn = 0.f, cblue = 0.f;

    for (int base = 0; base < Npix; base += 16) {
        float qd4 = 0.f, ld4 = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int px = base + 4 * j + e;
            const bool inb = px < Npix;
            const int pxc = inb ? px : Npix - 1;
            float d = bt.delta[rowN + pxc];
            const float sg = bt.error[rowN + pxc];
            const bool w = inb && svalid && (bt.mask[rowN + pxc] != 0);
            const bool blue = pxc < Nb;
            float A = 1.f, zd = 0.f, om = 0.f;
            if (blue) {
                BlueTerms t = blue_terms(bt.zabs[rowB + pxc], k);
                A = bt.A_blue ? bt.A_blue[rowB + pxc] : t.A;
                zd = t.zd;
                om = p.omega[pxc];
            }
            const float D = A * A * p.Psi[pxc] + om * zd + sg * sg;
            if (PREDICT) d = d - mu[pxc] * A;                       // QFA/model.py:166
            const float wD = w ? fast_rcp(D) : 0.f;
            d = w ? d : 0.f;
            const float wDA = wD * A;
            const float c2 = wDA * A, c3 = c2 * A, cb = wDA * d, cb2 = c2 * d;
            qd4 += wD * d * d;
            ld4 += w ? fast_log(D) : 0.f;
            cn += w ? 1.f : 0.f;
            cblue += (w && blue) ? 1.f : 0.f;
            const float *pfrow = PF + (size_t)px * C::NCP + sl;     // px < NpixPad always
#pragma unroll
            for (int t = 0; t < C::NFT; ++t) {
                const float fb = pfrow[16 * t];
                accb[t] = mfma4(cb, fb, accb[t]);
                accb2[t] = mfma4(cb2, fb, accb2[t]);
            }
#pragma unroll
            for (int t = 0; t < C::NT; ++t) {
                const float pb = pfrow[C::FW + 16 * t];
                accC[t] = mfma4(c2, pb, accC[t]);
                accT[t] = mfma4(c3, pb, accT[t]);
            }
        }
        qd += (double)qd4;
        ld += (double)ld4;
    }
    // C/D layout: col = lane&15, row = 4*(lane>>4) + r  -> spectrum s0 + 4j + r, column 16t + sl
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int ss = s0 + 4 * j + r;
        if (ss < B) {
            float *m = MOM + (size_t)ss * C::NMOM + sl;
#pragma unroll
            for (int t = 0; t < C::NT; ++t) {
                m[16 * t] = accC[t][r];
                m[C::MOM_T + 16 * t] = accT[t][r];
            }
#pragma unroll
            for (int t = 0; t < C::NFT; ++t) {
                m[C::MOM_B + 16 * t] = accb[t][r];
                m[C::MOM_B2 + 16 * t] = accb2[t][r];
            }
        }
    }
    qd += __shfl_xor(qd, 16); qd += __shfl_xor(qd, 32);
    ld += __shfl_xor(ld, 16); ld += __shfl_xor(ld, 32);
    cn += __shfl_xor(cn, 16); cn += __shfl_xor(cn, 32);
    cblue += __shfl_xor(cblue, 16); cblue += __shfl_xor(cblue, 32);
    if (j == 0 && svalid) {
        float *m = MOM + (size_t)s * C::NMOM + C::MOM_S;
        m[0] = (float)qd; m[1] = (float)ld; m[2] = cn; m[3] = cblue;
    }
}

// ------------------------------------------------------------------------------------------------
// k_solve.  KP lanes per spectrum, lane c holds column c (= row c) of the symmetric k x k
// matrices in registers; in-place Gauss-Jordan inversion in fp64, every step broadcasting the
// pivot column with wavefront shuffles.  The pivots are the squared Cholesky diagonal, so
// log det C = sum log(pivot) (finite where the reference's float32 det overflows, QFA/utils.py:54).
// ------------------------------------------------------------------------------------------------
template <int KP, bool PREDICT>
__global__ __launch_bounds__(256) void k_solve(const float *__restrict__ MOM, float *__restrict__ SOL,
                                               float *__restrict__ nll_out, int B, int Nh,
                                               float *__restrict__ hmean, float *__restrict__ hcov) {
    using C = Cfg<KP>;
    constexpr int G = 64 / KP;
    const int lane = threadIdx.x & 63;
    const int c = lane % KP;
    const int s = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * G + lane / KP;
    const bool valid = s < B;
    const float *mom = MOM + (size_t)(valid ? s : B - 1) * C::NMOM;

    double Cc[KP];
#pragma unroll
    for (int r = 0; r < KP; ++r) {
        const int a = r < c ? r : c, b = r < c ? c : r;
        Cc[r] = (double)mom[pair_index(a, b, KP)] + (r == c ? 1.0 : 0.0);
    }
    double logdet = 0.0;
#pragma unroll
    for (int jj = 0; jj < KP; ++jj) {
        double colj[KP];
#pragma unroll
        for (int i = 0; i < KP; ++i) colj[i] = __shfl(Cc[i], jj, KP);
        const double piv = colj[jj];
        logdet += log(piv);
        const double ip = 1.0 / piv;
        const double rjc = (c == jj) ? ip : Cc[jj] * ip;
#pragma unroll
        for (int i = 0; i < KP; ++i) {
            if (i != jj) Cc[i] = (c == jj) ? -colj[i] * ip : Cc[i] - colj[i] * rjc;
        }
        Cc[jj] = rjc;
    }
    // y = C^-1 b  (Cc[r] = Cinv[r][c] = Cinv[c][r])
    const double bc = (double)mom[C::MOM_B + c];
    double y = 0.0;
#pragma unroll
    for (int r = 0; r < KP; ++r) y += Cc[r] * __shfl(bc, r, KP);
    double quad = bc * y;
#pragma unroll
    for (int o = KP / 2; o >= 1; o >>= 1) quad += __shfl_xor(quad, o, KP);
    const float *sc = mom + C::MOM_S;
    const double nll = 0.5 * ((double)sc[0] - quad + (double)sc[2] * (double)QFA_LOG2PI + (double)sc[1] + logdet);
    if (valid && c == 0) nll_out[s] = (float)nll;

    float *sol = SOL + (size_t)(valid ? s : 0) * C::NSOL;
    if (valid) {
        sol[c] = (float)y;
#pragma unroll
        for (int r = 0; r < KP; ++r)
            if (r <= c) sol[C::SOL_CI + pair_index(r, c, KP)] = (float)(r == c ? Cc[r] : 2.0 * Cc[r]);
    }
    if (PREDICT) {
        if (valid && c < Nh) {
            hmean[(size_t)s * Nh + c] = (float)y;
#pragma unroll
            for (int r = 0; r < KP; ++r)
                if (r < Nh) hcov[((size_t)s * Nh + c) * Nh + r] = (float)Cc[r];
        }
        return;
    }
    // T column c (= row c), Z row c: Z[c][b] = sum_m Cinv[c][m] T[m][b]
    float Tc[KP];
#pragma unroll
    for (int r = 0; r < KP; ++r) {
        const int a = r < c ? r : c, b = r < c ? c : r;
        Tc[r] = mom[C::MOM_T + pair_index(a, b, KP)];
    }
    double Zr[KP];
#pragma unroll
    for (int b = 0; b < KP; ++b) Zr[b] = 0.0;
#pragma unroll
    for (int m = 0; m < KP; ++m) {
#pragma unroll
        for (int b = 0; b < KP; ++b) Zr[b] += Cc[m] * (double)__shfl(Tc[m], b, KP);
    }
    // p_c = b2_c - sum_m T[c][m] y_m
    double pc = (double)mom[C::MOM_B2 + c];
#pragma unroll
    for (int m = 0; m < KP; ++m) pc -= (double)Tc[m] * __shfl(y, m, KP);
    if (valid) {
#pragma unroll
        for (int b = 0; b < KP; ++b) sol[C::SOL_Z + c * KP + b] = (float)Zr[b];
        sol[C::SOL_P + c] = (float)pc;
    }
}

// ------------------------------------------------------------------------------------------------
// k_reduce_nll : accum scalars += {#spectra with an unmasked blue pixel, sum NLL, B}.  One block,
// fp64 partial sums, fixed order (deterministic).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_reduce_nll(const float *__restrict__ nll, const float *__restrict__ MOM,
                                                     int nmom, int mom_nblue, int B, float *__restrict__ scal) {
    __shared__ double sh[2][16];
    double a = 0.0, nb = 0.0;
    for (int s = threadIdx.x; s < B; s += blockDim.x) {
        a += (double)nll[s];
        nb += MOM[(size_t)s * nmom + mom_nblue] > 0.f ? 1.0 : 0.0;
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
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) { ta += sh[0][i]; tb += sh[1][i]; }
        scal[3] += (float)tb;
        scal[4] += (float)ta;
        scal[5] += (float)B;
    }
}

// ------------------------------------------------------------------------------------------------
// k_grads (pass 2).  One wave = 16 spectra, loops over 16-pixel tiles.  Lane (px = lane&15,
// g = lane>>4) owns pixel base+px of spectra s0 + 4g + r (r = 0..3):
//   stage 1  [fy | q] (16 spectra x 16 px) = [y | Cinv'] (regs, A operand) x PFT tile (B operand)
//            -> C/D layout col = px, row = 4g + r: the lane's own four elements
//   stage 2  u, diag(Sigma^-1), dG and the Psi / omega / tau0 / c0 / beta sums   (QFA/model.py:136-144)
//   stage 3  accF[px][b] += sum_{s,a} (wD A^2)_{s,px} f_{px,a} Z_s[a][b] + sum_s (A u)_{s,px} p_s[b]
//            A operand A[row = px][k = g] = beta_{s0+4g+r,px} f_{px,a} is lane-local; B operand
//            Z_{s0+4g+r}[a][b = lane&15] sits in registers for the whole pixel loop.
// gF = f * sumA - accF is formed in k_finalize (QFA/model.py:137 in low-rank form, App. A step 7).
// ------------------------------------------------------------------------------------------------
template <int KP>
__global__ __launch_bounds__(256) void k_grads(qfa_params_t p, qfa_batch_t bt, qfa_tau_t tau, int B, int Npix,
                                               int Nb, int Nh, int NpixPad, const float *__restrict__ PF,
                                               const float *__restrict__ PFT, const float *__restrict__ SOL,
                                               float *__restrict__ accum) {
    using C = Cfg<KP>;
    constexpr int KF = KP / 4, KQ = C::KK2 / 4;
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int s0 = tile * 16;
    if (s0 >= B) return;
    const DevConsts k = load_consts(p, tau);
    const int lo = lane & 15, g = lane >> 4;

    float *accF = accum;
    float *accA = accF + (size_t)Npix * Nh;
    float *accPsi = accA + Npix;
    float *accOm = accPsi + Npix;
    float *accCnt = accOm + Nb;
    float *accS = accCnt + Npix;

    // A operands of stage 1: spectrum s0+lo, k = 4t + g
    float yA[KF], qA[KQ];
    {
        const bool v = (s0 + lo) < B;
        const float *sol = SOL + (size_t)(v ? s0 + lo : 0) * C::NSOL;
#pragma unroll
        for (int t = 0; t < KF; ++t) yA[t] = v ? sol[4 * t + g] : 0.f;
#pragma unroll
        for (int t = 0; t < KQ; ++t) qA[t] = v ? sol[C::SOL_CI + 4 * t + g] : 0.f;
    }
    // B operands of stage 3: spectra s0+4g+r, column b = lo
    float Zr[4][KP], pr[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const bool v = (s0 + 4 * g + r) < B && lo < KP;
        const float *sol = SOL + (size_t)(v ? s0 + 4 * g + r : 0) * C::NSOL;
#pragma unroll
        for (int a = 0; a < KP; ++a) Zr[r][a] = v ? sol[C::SOL_Z + a * KP + lo] : 0.f;
        pr[r] = v ? sol[C::SOL_P + lo] : 0.f;
    }
    size_t rowN[4], rowB[4];
    bool sv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int s = s0 + 4 * g + r;
        sv[r] = s < B;
        rowN[r] = (size_t)(sv[r] ? s : B - 1) * Npix;
        rowB[r] = (size_t)(sv[r] ? s : B - 1) * Nb;
    }
    double s_tau0 = 0.0, s_c0 = 0.0, s_beta = 0.0;   // float32 per tile, float64 across tiles
    const int ntiles = NpixPad / 16;
    const int rot = (int)(((unsigned)tile * 2654435761u) % (unsigned)ntiles);   // de-phase the atomics

    for (int it = 0; it < ntiles; ++it) {
        int tt = it + rot;
        if (tt >= ntiles) tt -= ntiles;
        const int base = tt * 16;
        const int px = base + lo;
        const bool inb = px < Npix;
        const int pxc = inb ? px : Npix - 1;
        const bool blue = pxc < Nb;
        // ---- stage 1
        f32x4 afy = {0.f, 0.f, 0.f, 0.f}, aq = {0.f, 0.f, 0.f, 0.f};
        const float *pft = PFT + (size_t)g * NpixPad + px;
#pragma unroll
        for (int t = 0; t < KF; ++t) afy = mfma4(yA[t], pft[(size_t)(4 * t) * NpixPad], afy);
#pragma unroll
        for (int t = 0; t < KQ; ++t) aq = mfma4(qA[t], pft[(size_t)(KP + 4 * t) * NpixPad], aq);
        // ---- stage 2
        float f[KP];
        {
            const float4 *pfr = reinterpret_cast<const float4 *>(PF + (size_t)px * C::NCP);
#pragma unroll
            for (int a4 = 0; a4 < KP / 4; ++a4) {
                float4 v = pfr[a4];
                f[4 * a4] = v.x; f[4 * a4 + 1] = v.y; f[4 * a4 + 2] = v.z; f[4 * a4 + 3] = v.w;
            }
        }
        const float Psi = p.Psi[pxc];
        const float om = blue ? p.omega[pxc] : 0.f;
        float betaR[4], gamR[4];
        float gPsi = 0.f, gOm = 0.f, sA = 0.f, cnt = 0.f;
        float t_tau0 = 0.f, t_c0 = 0.f, t_beta = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float d = bt.delta[rowN[r] + pxc];
            const float sg = bt.error[rowN[r] + pxc];
            const bool w = inb && sv[r] && (bt.mask[rowN[r] + pxc] != 0);
            float A = 1.f, zd = 0.f, pw = 0.f, l2 = 0.f;
            if (blue) {
                BlueTerms t = blue_terms(bt.zabs[rowB[r] + pxc], k);
                A = bt.A_blue ? bt.A_blue[rowB[r] + pxc] : t.A;
                zd = t.zd; pw = t.pw; l2 = t.l2;
            }
            const float D = A * A * Psi + om * zd + sg * sg;
            const float wD = w ? fast_rcp(D) : 0.f;
            d = w ? d : 0.f;
            const float wDA = wD * A;
            const float u = wD * (d - A * afy[r]);                 // (Sigma^-1 delta)_i
            const float dS = wD - wDA * wDA * aq[r];               // diag(Sigma^-1)_i
            const float dG = 0.5f * (dS - u * u);                  // QFA/model.py:136,138
            gPsi += A * A * dG;                                    // :139
            gOm += dG * zd;                                        // :140
            const float root = 1.0f - k.tau0 * pw - k.c0;          // :141
            const float e = dG * (om * zd) * zd * 2.0f * root;
            t_tau0 -= e * pw;                                      // :142
            t_beta -= e * (k.tau0 * pw * (l2 * QFA_LN2));          // :143
            t_c0 -= e;                                             // :144
            cnt += w ? 1.f : 0.f;
            betaR[r] = wDA * A;
            sA += betaR[r] * A;
            gamR[r] = A * u;
        }
        s_tau0 += (double)t_tau0;
        s_c0 += (double)t_c0;
        s_beta += (double)t_beta;
        // ---- stage 3
        f32x4 aG = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int a = 0; a < KP; ++a) aG = mfma4(betaR[r] * f[a], Zr[r][a], aG);
            aG = mfma4(gamR[r], pr[r], aG);
        }
        // ---- flush.  aG: col = b = lo, row = 4g + rr -> pixel base + 4g + rr
        if (lo < Nh) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
                const int pp = base + 4 * g + rr;
                if (pp < Npix) atomicAdd(accF + (size_t)pp * Nh + lo, aG[rr]);
            }
        }
        gPsi += __shfl_xor(gPsi, 16); gPsi += __shfl_xor(gPsi, 32);
        gOm += __shfl_xor(gOm, 16); gOm += __shfl_xor(gOm, 32);
        sA += __shfl_xor(sA, 16); sA += __shfl_xor(sA, 32);
        cnt += __shfl_xor(cnt, 16); cnt += __shfl_xor(cnt, 32);
        if (g == 0 && inb) {
            atomicAdd(accPsi + px, gPsi);
            atomicAdd(accA + px, sA);
            atomicAdd(accCnt + px, cnt);
            if (blue) atomicAdd(accOm + px, gOm);
        }
    }
    for (int o = 32; o >= 1; o >>= 1) {
        s_tau0 += __shfl_xor(s_tau0, o);
        s_c0 += __shfl_xor(s_c0, o);
        s_beta += __shfl_xor(s_beta, o);
    }
    if (lane == 0) {
        atomicAdd(accS + 0, (float)s_tau0);
        atomicAdd(accS + 1, (float)s_c0);
        atomicAdd(accS + 2, (float)s_beta);
    }
}

// ------------------------------------------------------------------------------------------------
// k_predict_out : cont = F hmean + mu on ALL pixels, unc = sqrt(diag(F hcov F^T))  (QFA/model.py:180)
// -- stage 1 of k_grads with [hmean | hcov'] as the A operand.
// ------------------------------------------------------------------------------------------------
template <int KP>
__global__ __launch_bounds__(256) void k_predict_out(const float *__restrict__ mu, int B, int Npix, int NpixPad,
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
    for (int base = 0; base < Npix; base += 16) {
        const int px = base + lo;
        f32x4 afy = {0.f, 0.f, 0.f, 0.f}, aq = {0.f, 0.f, 0.f, 0.f};
        const float *pft = PFT + (size_t)g * NpixPad + px;
#pragma unroll
        for (int t = 0; t < KF; ++t) afy = mfma4(yA[t], pft[(size_t)(4 * t) * NpixPad], afy);
#pragma unroll
        for (int t = 0; t < KQ; ++t) aq = mfma4(qA[t], pft[(size_t)(KP + 4 * t) * NpixPad], aq);
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

// ------------------------------------------------------------------------------------------------
// k_finalize : grad = sum / count, elementwise, 0/0 = NaN (QFA/model.py:104); loss = sum NLL / B.
// ------------------------------------------------------------------------------------------------
__global__ void k_finalize(const float *__restrict__ accum, const float *__restrict__ F, int Npix, int Nb, int Nh,
                           int normalize, float *__restrict__ gF, float *__restrict__ gPsi, float *__restrict__ gOm,
                           float *__restrict__ gTau0, float *__restrict__ gC0, float *__restrict__ gBeta,
                           float *__restrict__ loss) {
    const float *accF = accum;
    const float *accA = accF + (size_t)Npix * Nh;
    const float *accPsi = accA + Npix;
    const float *accOm = accPsi + Npix;
    const float *accCnt = accOm + Nb;
    const float *accS = accCnt + Npix;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (size_t)Npix * Nh) {
        const int i = (int)(idx / Nh);
        const float v = F[idx] * accA[i] - accF[idx];
        gF[idx] = normalize ? v / accCnt[i] : v;
    }
    if (idx < (size_t)Npix) gPsi[idx] = normalize ? accPsi[idx] / accCnt[idx] : accPsi[idx];
    if (idx < (size_t)Nb) gOm[idx] = normalize ? accOm[idx] / accCnt[idx] : accOm[idx];
    if (idx == 0) {
        const float cs = normalize ? accS[3] : 1.f;
        *gTau0 = accS[0] / cs;
        *gC0 = accS[1] / cs;
        *gBeta = accS[2] / cs;
        *loss = normalize ? accS[4] / accS[5] : accS[4];
    }
}

// ------------------------------------------------------------------------------------------------
// k_adam_clip : Adam.update + clip for one tensor (QFA/optimizer.py:47-52, QFA/model.py:237-241)
// ------------------------------------------------------------------------------------------------
__global__ void k_adam_clip(const float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                            float *__restrict__ v, float *__restrict__ pout, size_t n, float lr, float b1, float b2,
                            float omb1, float omb2, float eps, float wd, float bc1, float bc2, float lo, float hi) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float pi = p[i];
    const float gi = g[i] + wd * pi;
    const float mi = omb1 * gi + b1 * m[i];
    const float vi = omb2 * gi * gi + b2 * v[i];
    m[i] = mi;
    v[i] = vi;
    float q = pi - lr * (mi / bc1) / (__fsqrt_rn(vi / bc2) + eps);
    if (lo <= hi) q = q < lo ? lo : (q > hi ? hi : q);   // NaN stays NaN, like torch.clip
    pout[i] = q;
}

__global__ void k_clip(const float *__restrict__ x, float *__restrict__ y, size_t n, float lo, float hi) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float q = x[i];
        y[i] = q < lo ? lo : (q > hi ? hi : q);
    }
}

// ------------------------------------------------------------------------------------------------
// k_smooth : edge-aware moving average along axis 0 (QFA/model.py:243-252)
// ------------------------------------------------------------------------------------------------
__global__ void k_smooth(const float *__restrict__ x, float *__restrict__ y, int n, int cols, int half) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)n * cols) return;
    const int i = (int)(idx / cols), c = (int)(idx % cols);
    const int lo = i - half < 0 ? 0 : i - half;
    const int hi = i + half + 1 > n ? n : i + half + 1;
    float acc = 0.f;
    for (int r = lo; r < hi; ++r) acc += x[(size_t)r * cols + c];
    y[idx] = acc / (float)(hi - lo);
}

// ------------------------------------------------------------------------------------------------
// elementwise optical-depth helpers (QFA/utils.py:57-92, 149-171)
// ------------------------------------------------------------------------------------------------
__global__ void k_tau(const float *__restrict__ z, float *__restrict__ out, size_t n, qfa_tau_t t) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = t.amp * powf((1.0f + z[i]) * t.scale, t.expo) + t.offset;
}
__global__ void k_tauhi(const float *__restrict__ z, const float *tau0, const float *beta, float *__restrict__ out,
                        size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = *tau0 * powf(1.0f + z[i], *beta);
}
__global__ void k_omega_func(const float *__restrict__ z, const float *tau0, const float *beta, const float *c0,
                             float *__restrict__ out, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const float r = 1.0f - *c0 - expf(-(*tau0 * powf(1.0f + z[i], *beta)));
        out[i] = r * r;
    }
}

// ------------------------------------------------------------------------------------------------
// Woodbury utilities for one (n,k) M and (n,) D (QFA/utils.py:12-54): small, generic, fp64 core.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_wood_core(const float *__restrict__ M, const float *__restrict__ D, int n,
                                                    int k, double *__restrict__ core /* k*k + 1 */) {
    extern __shared__ double sh[];   // k*k
    const int t = threadIdx.x;
    if (t < k * k) {
        const int a = t / k, b = t % k;
        double acc = a == b ? 1.0 : 0.0;
        for (int i = 0; i < n; ++i) acc += (double)M[(size_t)i * k + a] * (double)M[(size_t)i * k + b] / (double)D[i];
        sh[t] = acc;
    }
    __syncthreads();
    // sum log D on all threads
    double ld = 0.0;
    for (int i = t; i < n; i += blockDim.x) ld += log((double)D[i]);
    for (int o = 32; o >= 1; o >>= 1) ld += __shfl_xor(ld, o);
    __shared__ double shl[16];
    if ((t & 63) == 0) shl[t >> 6] = ld;
    __syncthreads();
    double logdet = 0.0;
    for (int jj = 0; jj < k; ++jj) {           // in-place Gauss-Jordan, threads over (i, c)
        const double piv = sh[jj * k + jj];
        __syncthreads();
        double nv = 0.0;
        const int i = t / k, c = t % k;
        if (t < k * k) {
            const double ip = 1.0 / piv;
            const double aij = sh[i * k + jj], ajc = sh[jj * k + c];
            if (i == jj) nv = (c == jj) ? ip : ajc * ip;
            else nv = (c == jj) ? -aij * ip : sh[t] - aij * ajc * ip;
        }
        logdet += log(piv);
        __syncthreads();
        if (t < k * k) sh[t] = nv;
        __syncthreads();
    }
    if (t < k * k) core[t] = sh[t];
    if (t == 0) {
        double s = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += shl[i];
        core[k * k] = s + logdet;
    }
}

__global__ void k_wood_inv(const float *__restrict__ M, const float *__restrict__ D, int n, int k,
                           const double *__restrict__ core, float *__restrict__ inv, float *__restrict__ logdet) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx == 0 && logdet) *logdet = (float)core[k * k];
    if (!inv || idx >= (size_t)n * n) return;
    const int i = (int)(idx / n), j = (int)(idx % n);
    double acc = 0.0;
    for (int a = 0; a < k; ++a) {
        double t = 0.0;
        for (int b = 0; b < k; ++b) t += core[a * k + b] * (double)M[(size_t)j * k + b];
        acc += (double)M[(size_t)i * k + a] * t;
    }
    const double di = 1.0 / (double)D[i], dj = 1.0 / (double)D[j];
    inv[idx] = (float)((i == j ? di : 0.0) - di * acc * dj);
}

// ================================================================================================
// C ABI
// ================================================================================================
namespace {

const double kLymanF[30] = {4.1620e-01, 7.9140e-02, 2.9010e-02, 1.3950e-02, 7.8030e-03, 4.8160e-03, 3.1850e-03,
                            2.2170e-03, 1.6060e-03, 1.2010e-03, 9.2190e-04, 7.2310e-04, 5.7770e-04, 4.6890e-04,
                            3.8580e-04, 3.2120e-04, 2.7030e-04, 2.2970e-04, 1.9680e-04, 1.6990e-04, 1.4770e-04,
                            1.2930e-04, 1.1370e-04, 1.0060e-04, 8.9360e-05, 7.9780e-05, 7.1480e-05, 6.4350e-05,
                            5.8120e-05, 5.2640e-05};
const double kLymanLam[30] = {1215.6701, 1025.7222, 972.5367, 949.7430, 937.8034, 930.7482, 926.2256, 923.1503,
                              920.9630,  919.3513,  918.1293, 917.1805, 916.4291, 915.8238, 915.3289, 914.9192,
                              914.5762,  914.2861,  914.0385, 913.8256, 913.6411, 913.4803, 913.3391, 913.2146,
                              913.1042,  913.0059,  912.9179, 912.8389, 912.7676, 912.7032};

inline int kp_for(int Nh) { return Nh <= 8 ? 8 : (Nh <= 16 ? 16 : 32); }

struct Layout {
    int KP, NpixPad, Bpad, NCP, NCT, NMOM, NSOL, MOM_S;
    size_t oPF, oPFT, oMOM, oSOL, oNLL, total;   // float offsets
};

template <int KP>
Layout make_layout_t(int B, int Npix) {
    using C = Cfg<KP>;
    Layout L;
    L.KP = KP;
    L.NpixPad = round_up(Npix, 16);
    L.Bpad = round_up(B, 16);
    L.NCP = C::NCP; L.NCT = C::NCT; L.NMOM = C::NMOM; L.NSOL = C::NSOL; L.MOM_S = C::MOM_S;
    size_t o = 0;
    auto take = [&](size_t n) { size_t r = o; o += (n + 63) / 64 * 64; return r; };
    L.oPF = take((size_t)L.NpixPad * C::NCP);
    L.oPFT = take((size_t)C::NCT * L.NpixPad);
    L.oMOM = take((size_t)L.Bpad * C::NMOM);
    L.oSOL = take((size_t)L.Bpad * C::NSOL);
    L.oNLL = take((size_t)L.Bpad);
    L.total = o;
    return L;
}

Layout make_layout(int B, int Npix, int Nh) {
    switch (kp_for(Nh)) {
        case 8: return make_layout_t<8>(B, Npix);
        case 16: return make_layout_t<16>(B, Npix);
        default: return make_layout_t<32>(B, Npix);
    }
}

inline int check_shape(int B, int Npix, int Nb, int Nh) {
    if (B < 1 || Npix < 1 || Nb < 0 || Nb > Npix || Nh < 1 || Nh > 16) return QFA_E_SIZE;
    return 0;
}

inline int hip_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? 0 : (int)e;
}

template <int KP>
int run_nll_grad(const qfa_params_t &p, const qfa_batch_t &b, const qfa_tau_t &tau, int B, int Npix, int Nb, int Nh,
                 float *nll, float *accum, float *ws, hipStream_t st, void *const *events) {
    using C = Cfg<KP>;
    const Layout L = make_layout_t<KP>(B, Npix);
    float *PF = ws + L.oPF, *PFT = ws + L.oPFT, *MOM = ws + L.oMOM, *SOL = ws + L.oSOL;
    float *nllbuf = nll ? nll : ws + L.oNLL;
    const size_t accS = (size_t)Npix * Nh + 3 * (size_t)Npix + Nb;
    auto mark = [&](int i) {
        if (events && events[i]) (void)hipEventRecord((hipEvent_t)events[i], st);
    };
    mark(0);
    {
        dim3 blk(64, 4);
        k_prep_pf<KP><<<(L.NpixPad + 3) / 4, blk, 0, st>>>(p.F, Npix, Nh, L.NpixPad, PF, PFT);
    }
    mark(1);
    const int tiles = (B + 15) / 16;
    k_moments<KP, false><<<(tiles + 3) / 4, 256, 0, st>>>(p, b, tau, nullptr, B, Npix, Nb, PF, MOM);
    mark(2);
    constexpr int G = 64 / KP;
    k_solve<KP, false><<<(B + 4 * G - 1) / (4 * G), 256, 0, st>>>(MOM, SOL, nllbuf, B, Nh, nullptr, nullptr);
    k_reduce_nll<<<1, 1024, 0, st>>>(nllbuf, MOM, C::NMOM, C::MOM_S + 3, B, accum + accS);
    mark(3);
    k_grads<KP><<<(tiles + 3) / 4, 256, 0, st>>>(p, b, tau, B, Npix, Nb, Nh, L.NpixPad, PF, PFT, SOL, accum);
    mark(4);
    return hip_status();
}

template <int KP>
int run_predict(const qfa_params_t &p, const float *mu, const qfa_batch_t &b, const qfa_tau_t &tau, int B, int Npix,
                int Nb, int Nh, float *ll, float *hmean, float *hcov, float *cont, float *unc, float *ws,
                hipStream_t st) {
    const Layout L = make_layout_t<KP>(B, Npix);
    float *PF = ws + L.oPF, *PFT = ws + L.oPFT, *MOM = ws + L.oMOM, *SOL = ws + L.oSOL;
    {
        dim3 blk(64, 4);
        k_prep_pf<KP><<<(L.NpixPad + 3) / 4, blk, 0, st>>>(p.F, Npix, Nh, L.NpixPad, PF, PFT);
    }
    const int tiles = (B + 15) / 16;
    k_moments<KP, true><<<(tiles + 3) / 4, 256, 0, st>>>(p, b, tau, mu, B, Npix, Nb, PF, MOM);
    constexpr int G = 64 / KP;
    k_solve<KP, true><<<(B + 4 * G - 1) / (4 * G), 256, 0, st>>>(MOM, SOL, ll, B, Nh, hmean, hcov);
    k_predict_out<KP><<<(tiles + 3) / 4, 256, 0, st>>>(mu, B, Npix, L.NpixPad, PFT, SOL, cont, unc);
    return hip_status();
}

}  // namespace

extern "C" {

int qfa_abi_version(void) { return QFA_ABI_VERSION; }

int qfa_tau_model(int which, int series, qfa_tau_t *out) {
    if (!out) return QFA_E_NULL;
    if (series < 1 || series > 30) return QFA_E_TAU;
    const double coeff = kLymanLam[series - 1] * kLymanF[series - 1] / (kLymanLam[0] * kLymanF[0]);
    double amp, scale, expo, off;
    switch (which) {
        case QFA_TAU_BECKER: amp = 0.751; scale = 1.0 / 4.5; expo = 2.90; off = -0.132; break;
        case QFA_TAU_FG: amp = 0.0018; scale = 1.0; expo = 3.92; off = 0.0; break;
        case QFA_TAU_KAMBLE: amp = 5.54e-3; scale = 1.0; expo = 3.182; off = 0.0; break;
        case QFA_TAU_MOCK: amp = 0.2231435513142097; scale = 1.0 / 3.25; expo = 3.2; off = 0.0; break;
        default: return QFA_E_TAU;
    }
    out->amp = (float)(amp * coeff);
    out->scale = (float)scale;
    out->expo = (float)expo;
    out->offset = (float)(off * coeff);
    return 0;
}

size_t qfa_workspace_bytes(int B, int Npix, int Nh) {
    if (B < 1 || Npix < 1 || Nh < 1 || Nh > 32) return 0;
    return make_layout(B, Npix, Nh).total * sizeof(float);
}

size_t qfa_accum_floats(int Npix, int Nb, int Nh) {
    return (size_t)Npix * Nh + 3 * (size_t)Npix + (size_t)Nb + 8;
}

int qfa_nll_grad_events_f32(const qfa_params_t *p, const qfa_batch_t *b, const qfa_tau_t *tau, int B, int Npix,
                            int Nb, int Nh, float *nll, float *accum, void *workspace, size_t workspace_bytes,
                            void *stream, void *const *events);

int qfa_nll_grad_f32(const qfa_params_t *p, const qfa_batch_t *b, const qfa_tau_t *tau, int B, int Npix, int Nb,
                     int Nh, float *nll, float *accum, void *workspace, size_t workspace_bytes, void *stream) {
    return qfa_nll_grad_events_f32(p, b, tau, B, Npix, Nb, Nh, nll, accum, workspace, workspace_bytes, stream, nullptr);
}

int qfa_nll_grad_events_f32(const qfa_params_t *p, const qfa_batch_t *b, const qfa_tau_t *tau, int B, int Npix,
                            int Nb, int Nh, float *nll, float *accum, void *workspace, size_t workspace_bytes,
                            void *stream, void *const *events) {
    if (!p || !b || !tau || !accum || !workspace) return QFA_E_NULL;
    if (!p->F || !p->Psi || !p->tau0 || !p->c0 || !p->beta || (Nb > 0 && !p->omega)) return QFA_E_NULL;
    if (!b->delta || !b->error || !b->mask || (Nb > 0 && !b->zabs)) return QFA_E_NULL;
    if (int e = check_shape(B, Npix, Nb, Nh)) return e;
    if (workspace_bytes < qfa_workspace_bytes(B, Npix, Nh)) return QFA_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float *ws = (float *)workspace;
    if (kp_for(Nh) == 8) return run_nll_grad<8>(*p, *b, *tau, B, Npix, Nb, Nh, nll, accum, ws, st, events);
    return run_nll_grad<16>(*p, *b, *tau, B, Npix, Nb, Nh, nll, accum, ws, st, events);
}

int qfa_finalize_grads_f32(const float *accum, const float *F, int Npix, int Nb, int Nh, int normalize, float *gF,
                           float *gPsi, float *gOmega, float *gTau0, float *gC0, float *gBeta, float *loss,
                           void *stream) {
    if (!accum || !F || !gF || !gPsi || !gTau0 || !gC0 || !gBeta || !loss || (Nb > 0 && !gOmega)) return QFA_E_NULL;
    if (int e = check_shape(1, Npix, Nb, Nh)) return e;
    const size_t n = (size_t)Npix * Nh;
    k_finalize<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(accum, F, Npix, Nb, Nh, normalize, gF, gPsi,
                                                                            gOmega, gTau0, gC0, gBeta, loss);
    return hip_status();
}

int qfa_predict_f32(const qfa_params_t *p, const float *mu, const qfa_batch_t *b, const qfa_tau_t *tau, int B,
                    int Npix, int Nb, int Nh, float *ll, float *hmean, float *hcov, float *cont, float *unc,
                    void *workspace, size_t workspace_bytes, void *stream) {
    if (!p || !b || !tau || !mu || !ll || !hmean || !hcov || !cont || !unc || !workspace) return QFA_E_NULL;
    if (!p->F || !p->Psi || !p->tau0 || !p->c0 || !p->beta || (Nb > 0 && !p->omega)) return QFA_E_NULL;
    if (!b->delta || !b->error || !b->mask || (Nb > 0 && !b->zabs)) return QFA_E_NULL;
    if (int e = check_shape(B, Npix, Nb, Nh)) return e;
    if (workspace_bytes < qfa_workspace_bytes(B, Npix, Nh)) return QFA_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    float *ws = (float *)workspace;
    if (kp_for(Nh) == 8) return run_predict<8>(*p, mu, *b, *tau, B, Npix, Nb, Nh, ll, hmean, hcov, cont, unc, ws, st);
    return run_predict<16>(*p, mu, *b, *tau, B, Npix, Nb, Nh, ll, hmean, hcov, cont, unc, ws, st);
}

int qfa_adam_clip_f32(const float *p, const float *g, float *m, float *v, float *p_out, size_t n, double lr, double b1,
                      double b2, double eps, double wd, int i, float lo, float hi, void *stream) {
    if (!p || !g || !m || !v || !p_out) return QFA_E_NULL;
    if (n == 0) return 0;
    if (i < 0) return QFA_E_SIZE;
    // The reference mixes Python floats (double) with float32 tensors: every scalar below is
    // formed in double and rounded to float32 once, exactly where torch would round it.
    const float bc1 = (float)(1.0 - pow(b1, (double)(i + 1))), bc2 = (float)(1.0 - pow(b2, (double)(i + 1)));
    k_adam_clip<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(
        p, g, m, v, p_out, n, (float)lr, (float)b1, (float)b2, (float)(1.0 - b1), (float)(1.0 - b2), (float)eps,
        (float)wd, bc1, bc2, lo, hi);
    return hip_status();
}

int qfa_clip_f32(const float *x, float *y, size_t n, float lo, float hi, void *stream) {
    if (!x || !y) return QFA_E_NULL;
    if (n == 0) return 0;
    k_clip<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, y, n, lo, hi);
    return hip_status();
}

int qfa_smooth_f32(const float *x, float *y, int n, int cols, int half, void *stream) {
    if (!x || !y) return QFA_E_NULL;
    if (n < 1 || cols < 1 || half < 0) return QFA_E_SIZE;
    const size_t tot = (size_t)n * cols;
    k_smooth<<<(unsigned)((tot + 255) / 256), 256, 0, (hipStream_t)stream>>>(x, y, n, cols, half);
    return hip_status();
}

int qfa_tau_f32(const float *z, float *out, size_t n, const qfa_tau_t *tau, void *stream) {
    if (!z || !out || !tau) return QFA_E_NULL;
    if (n == 0) return 0;
    k_tau<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(z, out, n, *tau);
    return hip_status();
}

int qfa_tauhi_f32(const float *z, const float *tau0, const float *beta, float *out, size_t n, void *stream) {
    if (!z || !out || !tau0 || !beta) return QFA_E_NULL;
    if (n == 0) return 0;
    k_tauhi<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(z, tau0, beta, out, n);
    return hip_status();
}

int qfa_omega_func_f32(const float *z, const float *tau0, const float *beta, const float *c0, float *out, size_t n,
                       void *stream) {
    if (!z || !out || !tau0 || !beta || !c0) return QFA_E_NULL;
    if (n == 0) return 0;
    k_omega_func<<<(unsigned)((n + 255) / 256), 256, 0, (hipStream_t)stream>>>(z, tau0, beta, c0, out, n);
    return hip_status();
}

int qfa_woodbury_f32(const float *M, const float *D, int n, int k, float *inv, float *logdet, void *workspace,
                     size_t workspace_bytes, void *stream) {
    if (!M || !D || !workspace || (!inv && !logdet)) return QFA_E_NULL;
    if (n < 1 || k < 1 || k > 32) return QFA_E_SIZE;
    if (workspace_bytes < (size_t)(k * k + 1) * sizeof(double)) return QFA_E_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    double *core = (double *)workspace;
    k_wood_core<<<1, 1024, (size_t)k * k * sizeof(double), st>>>(M, D, n, k, core);
    const size_t tot = inv ? (size_t)n * n : 1;
    k_wood_inv<<<(unsigned)((tot + 255) / 256), 256, 0, st>>>(M, D, n, k, core, inv, logdet);
    return hip_status();
}

}  // extern "C"

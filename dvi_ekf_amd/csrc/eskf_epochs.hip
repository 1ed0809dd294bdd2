// eskf_epochs.hip -- the long-horizon kernel: a whole run of epochs (N IMU samples, then one camera update; the time loop
// of Filter.run / run_one_epoch / propagate_imu, dvi_ekf/filter/Filter.py:144-217) in ONE launch, for any N >= 0.
//
// A block owns NI = 2 NC filter instances (fp32 also 4 NC: two pairs per covariance wavefront, Lay::PP) for the whole run.  Its wavefronts split the work of the reference's
// Filter.propagate (Filter.py:219-230) and Filter.update (351-395) the way the data depend on each other:
//
//   RA   (wavefront 0, lane = instance)  the recurrence of _predict_nominal that carries attitude, position, velocity
//        (Filter.py:232-247, equations.py:44-47,72-85); per predict it leaves q_old, om_old, acc_old, dt behind
//   RB   (wavefront 1, lane = instance)  camera attitude and notch states (equations.py:50,80-86); per predict the
//        post-predict notch state and sin / cos of the post-predict phi; at the end of an epoch the residual (364-375)
//   JA   (wavefront 2, lane = (predict, instance))  Jacobian blocks A, Theta, Bm, Cm of _predict_error /
//   JB   (wavefront 3, lane = (predict, instance))  _cam_error_jacobian (249-342): Dm, Em -- of up to PPP predicts at once
//   C_w  (wavefronts 4 .. 4 + NC - 1, 32 lanes per instance)  _predict_error_covariance (344-349), the update and the
//        reset (351-390) on the rows of P -- which stay in the wavefront's registers from the first epoch to the last
//
// R -> J -> C is a pipeline through LDS: an `aux` ring (what only a recurrence can know, 15 words per predict and
// instance), a ring of 67-word Jacobian records, a mailbox per instance (error state and status word of the update on its
// way back to the recurrences, the residual on its way to the update, the epoch's probe trigonometry).  Nobody waits at a
// block barrier: every hand-over is a monotonic counter in LDS (predicts / epochs done) that the reader polls; the
// recurrence of predict i + k runs while the covariance wavefronts work on predict i, and covariance predict i only
// needs record i.  The only place the pipeline drains is the update: the next epoch's recurrences start from the
// corrected state (State.__add__, state.py:46-60).
//
// The arithmetic is that of the epoch-by-epoch path (eskf_nominal_kernel + eskf_jac_kernel + eskf_cov_rows_kernel in
// eskf_kernels.hip): the same device functions where there are any, the same expressions in the same order elsewhere;
// tests/test_gpu_epochs_kernel.py compares the two bit for bit.  What this kernel removes: three launches per camera
// frame, the Jacobian records' round trip through memory (2 x 600 B per predict and instance), the covariance's round
// trip per epoch (2 x 2.4 KB), and the serialisation of recurrences -> blocks -> covariance within a frame.
//
// Compiled as a translation unit of its own (Makefile) without MachineLICM, like the one-launch step: the bodies below sit
// in loops over epochs and predicts, and the pass would hoist every constant they materialise in front of them.
#define ESKF_TU_EPOCHS 1
#include "eskf_kernels.hip"

namespace eskf {
namespace ep {

// ---- LDS layout (words of T; the counters are ints behind them) ------------------------------------------------
constexpr int IMG = 600;                 // exchange image of one instance (cov_rows_body<WAVE>: 24 x 25 transposition)
constexpr int RW = JB_CORE;              // one Jacobian record: the 67 block words in the member order of Jac<T>
constexpr int AW = 17;                   // one aux entry (odd stride), words:
constexpr int AX_QO = 0, AX_OM = 4, AX_ACC = 7, AX_DT = 10;         // RA: q_old(k) om_old(k) acc_old(k) dt(k)
constexpr int AX_N0 = 11, AX_N1 = 12, AX_SP = 13, AX_CP = 14;       // RB: notch angle / rate after predict k, sin / cos phi+
constexpr int MW = 53;                   // mailbox of one instance (odd stride), words:
constexpr int MB_E = 0, MB_ST = 24;      // C -> R, J: error state of the update, status word (as a T word)
constexpr int MB_RES = 25;               // RB -> C: residual 7 + non-finite flag
constexpr int MB_PC = 33;                // RA -> RB: p_cam after the epoch's last predict
constexpr int MB_T0 = 36;                // RA -> RB: s1 c1 s2 c2 at the dofs the epoch's first predict sees
constexpr int MB_T1 = 40;                // RA -> J: s1 c1 s2 c2 s3 c3 at the post-predict dofs (frozen ones zeroed)
constexpr int MB_DOFS = 46;              // RA -> JA: the post-predict dofs
// counters
constexpr int F_RA = 0, F_RB = 1;        // aux entries published (predicts of the run)
constexpr int F_JA = 2, F_JB = 3;        // records completed
constexpr int F_RA_CONST = 4;            // epochs whose trigonometry / dofs RA has published
constexpr int F_RA_EP = 5, F_RB_EP = 6;  // epochs whose p_cam / residual are in the mailboxes
constexpr int F_LOADED = 8;              // [NC] records consumer wavefront w has taken into its registers
constexpr int F_CEPOCH = 24;             // [NC] epochs whose update wavefront w has finished (error state in the mailbox)
constexpr int NFLAGS = 40;

// NC covariance wavefronts, each with PP pairs of instances (PP = 2: fp32 only -- two rows of 24 words per lane; the
// producers then serve 4 NC instances, with twice the lanes of their wavefronts at work)
template <typename T, int NC_, int PP_ = 1>
struct Lay {
    static constexpr int NC = NC_, PP = PP_;
    static constexpr int NI = 2 * NC * PP;
    static constexpr int NIP = NI <= 4 ? 4 : NI <= 8 ? 8 : NI <= 16 ? 16 : NI <= 32 ? 32 : 64;     // lanes per predict in the J wavefronts
    static constexpr int PPP = 64 / NIP < 4 ? 64 / NIP : 4;                        // predicts per J pass
    // ring depths: what fits next to the images (24 instances in fp64, 48 in fp32: 115 KB of images)
    static constexpr int D = PP > 1 ? (NC > 10 ? 2 : NC > 8 ? 4 : 6)                                                        // records
                                    : NC > 10 ? (sizeof(T) == 8 ? 2 : 4) : NC > 8 ? 4 : NC > 4 ? (sizeof(T) == 8 ? 4 : 6) : 8;
    static constexpr int DA = PP > 1 ? (NC > 10 ? 2 : NC > 8 ? 4 : 8) : NC > 10 ? (sizeof(T) == 8 ? 2 : 4) : NC > 8 ? 4 : 8;    // aux entries
    static_assert(NI <= 64 && (PP == 1 || sizeof(T) == 4), "lane = instance in the recurrences; two pairs per wavefront in fp32 only");
    static constexpr int O_IMG = 0;
    static constexpr int O_RING = O_IMG + NI * IMG;
    static constexpr int O_AUX = O_RING + D * NI * RW;
    static constexpr int O_MBOX = O_AUX + DA * NI * AW;
    static constexpr int O_FLAGS = O_MBOX + NI * MW;            // (words of T; ints from here)
    static constexpr size_t BYTES = (size_t)O_FLAGS * sizeof(T) + NFLAGS * sizeof(int);
};

// ---- hand-over primitives -----------------------------------------------------------------------------------------
// A counter in LDS, read by every lane of the (wave-uniform) reader at the same address.  LDS executes the instructions
// of all wavefronts of the CU in one order: a reader that sees the counter sees every LDS write its writer issued before
// it.  What is left to do is to keep the COMPILER from moving accesses across (the asm statements).
typedef int __attribute__((address_space(3))) * lds_int_p;
__device__ __forceinline__ int peek(const int* p) {
    const int v = __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __builtin_amdgcn_readfirstlane(v);
}
#ifdef ESKF_X_EPSTAMPS
#define EP_WAIT_BEGIN() { const unsigned long long tw0_ = __builtin_amdgcn_s_memtime();
#define EP_WAIT_END(SLOT) wait_cycles[SLOT] += __builtin_amdgcn_s_memtime() - tw0_; }
#else
#define EP_WAIT_BEGIN() {
#define EP_WAIT_END(SLOT) }
#endif
__device__ __forceinline__ void wait_ge(const int* p, int need) {
    while (peek(p) < need) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
// every counter of counters[0..n) >= need (lane l looks at counter l)
__device__ __forceinline__ bool all_ge(const int* counters, int n, int need, int lane) {
    const int v = __hip_atomic_load(counters + (lane < n ? lane : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __builtin_amdgcn_ballot_w64(v < need) == 0ull;
}
// some counter of counters[0..n) >= need
__device__ __forceinline__ bool any_ge(const int* counters, int n, int need, int lane) {
    const int v = __hip_atomic_load(counters + (lane < n ? lane : 0), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __builtin_amdgcn_ballot_w64(v >= need) != 0ull;
}
__device__ __forceinline__ void wait_all_ge(const int* counters, int n, int need, int lane) {
    while (!all_ge(counters, n, need, lane)) __builtin_amdgcn_s_sleep(1);
    asm volatile("" ::: "memory");
}
__device__ __forceinline__ void publish(int* p, int v) {
    asm volatile("" ::: "memory");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    asm volatile("" ::: "memory");
}

// number of predicts of epoch e
__device__ __forceinline__ int n_pred_of(const EpochsArgs& A, int e) {
    if (A.n_pred == nullptr) return A.n_uniform;
    typedef const int32_t __attribute__((address_space(4))) * cint_p;      // (read-only for the launch: scalar loads)
    return ((cint_p)(uintptr_t)A.n_pred)[e];
}

// IMU sample g of the run for instance b (zeros beyond the run's last one: assigned on every path)
template <typename T>
__device__ __forceinline__ void load_sample(const StepArgs& a, int g, int g_end, int64_t b, T& dt, T (&om)[3], T (&acc)[3]) {
    if (g < g_end) {
        const int64_t so = a.imu_stride ? ((int64_t)g * a.B + b) * 3 : (int64_t)g * 3;
        dt = reinterpret_cast<const T*>(a.dt)[a.dt_stride ? (int64_t)g * a.B + b : g];
#pragma unroll
        for (int i = 0; i < 3; ++i) { om[i] = reinterpret_cast<const T*>(a.om)[so + i]; acc[i] = reinterpret_cast<const T*>(a.acc)[so + i]; }
    } else {
        dt = T(0);
#pragma unroll
        for (int i = 0; i < 3; ++i) { om[i] = T(0); acc[i] = T(0); }
    }
}

// the pending error state of an instance: from memory at the start of the run (what an earlier call left behind), from
// the mailbox afterwards (what this run's own update of the epoch before has put there)
template <typename T>
struct Pending {
    T src[24];          // (the words a role uses: the others are never loaded)
    int status;
    bool inject;
};
// FIRST / COUNT...: the word ranges the calling role needs (up to three)
template <typename T, int F0, int N0, int F1, int N1, int F2 = 0, int N2 = 0, int F3 = 0, int N3 = 0>
__device__ __forceinline__ Pending<T> pending_of(const StepArgs& a, int e, int64_t b, const T* mb, const int* flags, int nca, int lane) {
    Pending<T> pd;
    auto take = [&](const T* src) {
#pragma unroll
        for (int i = 0; i < N0; ++i) pd.src[F0 + i] = src[F0 + i];
#pragma unroll
        for (int i = 0; i < N1; ++i) pd.src[F1 + i] = src[F1 + i];
#pragma unroll
        for (int i = 0; i < N2; ++i) pd.src[F2 + i] = src[F2 + i];
#pragma unroll
        for (int i = 0; i < N3; ++i) pd.src[F3 + i] = src[F3 + i];
    };
    if (e == 0) {
        pd.status = a.pending_hint ? a.status_ws[b] : 0;
        const bool was = (pd.status & DVI_STATUS_PENDING) != 0;
        pd.inject = was && !a.drop_pending && !(pd.status & DVI_STATUS_SINGULAR);
        if (pd.inject) take(reinterpret_cast<const T*>(a.e) + b * 24);
    } else {
        wait_all_ge(flags + F_CEPOCH, nca, e, lane);
        pd.status = word_as_status(mb[MB_ST]);
        pd.inject = (pd.status & DVI_STATUS_PENDING) != 0 && !(pd.status & DVI_STATUS_SINGULAR);
        take(mb + MB_E);
    }
    return pd;
}

// ---- RA: position, velocity, attitude (the arithmetic of eskf_nominal_kernel, role 0) ----------------------------
template <typename T, typename L>
__device__ __forceinline__ void role_ra(const EpochsArgs& A, const DevParams<T>& prm, T* lds, const int lane, const int64_t b0, const int ni,
                                        const int nca, unsigned long long* wait_cycles) {
    constexpr int NI = L::NI;
    if (lane >= NI) return;
    const StepArgs& a = A.a;
    T* aux = lds + L::O_AUX;
    T* mb = lds + L::O_MBOX + lane * MW;
    int* flags = reinterpret_cast<int*>(lds + L::O_FLAGS);
    const bool live = lane < ni;
    const int64_t b = b0 + (live ? lane : ni - 1);      // (a ragged block's tail lanes shadow its last instance, write nothing to memory)
    const int64_t B = a.B;
    const T* xi = reinterpret_cast<const T*>(a.xs_src);
    auto X = [&](int w) { return xi[(int64_t)w * B + b]; };
    bool zeroed[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) zeroed[i] = !prm.hold_frozen && ((prm.frozen_mask >> i) & 1);   // Filter.py:243-245
    T p[3], v[3], q[4], pc[3], om_o[3], acc_o[3], qo[4], dofs[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) { p[i] = X(i); v[i] = X(3 + i); pc[i] = X(19 + i); om_o[i] = X(26 + i); acc_o[i] = X(29 + i); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { q[i] = X(6 + i); qo[i] = X(32 + i); }
#pragma unroll
    for (int i = 0; i < 6; ++i) dofs[i] = X(10 + i);
    // sines / cosines of q1..q3 together with the arguments they belong to: re-evaluated where an injection has moved a
    // joint (same argument, same library call: the bits of an evaluation per epoch)
    T targ[3] = {T(NAN), T(NAN), T(NAN)}, ts[3] = {T(0), T(0), T(0)}, tcs[3] = {T(1), T(1), T(1)};
    int g = 0;
    for (int e = 0; e < A.n_epochs; ++e) {
        const int n = n_pred_of(A, e);
        const int g_end = g + n;
        // measured camera position of this epoch and the first sample: requested before the wait for the update
        T gpos[3];
        {
            const T* gp = reinterpret_cast<const T*>(a.pos) + ((int64_t)e * (a.meas_stride ? B : 1) + (a.meas_stride ? b : 0)) * 3;
#pragma unroll
            for (int i = 0; i < 3; ++i) gpos[i] = gp[i];
        }
        T dt, om_n[3], acc_n[3];
        load_sample(a, g, g_end, b, dt, om_n, acc_n);
        Pending<T> pd;
        EP_WAIT_BEGIN()
        pd = pending_of<T, 0, 15, 18, 3>(a, e, b, mb, flags, nca, lane);
        EP_WAIT_END(0)
        if (pd.inject) {                                // state.py:46-60,105-129
            T e15[15], e1821[3];
#pragma unroll
            for (int i = 0; i < 15; ++i) e15[i] = pd.src[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) e1821[i] = pd.src[18 + i];
            T th[3] = {e15[6], e15[7], e15[8]}, dq[4], qn[4];
            quat_about_axis(norm3(th), th, dq);
            quat_mul(q, dq, qn);
            quat_nrm(qn);
#pragma unroll
            for (int i = 0; i < 3; ++i) { p[i] += e15[i]; v[i] += e15[3 + i]; pc[i] += e1821[i]; }
#pragma unroll
            for (int i = 0; i < 6; ++i) dofs[i] += e15[9 + i];
#pragma unroll
            for (int i = 0; i < 4; ++i) q[i] = qn[i];
        }
        // probe: p_CB at the dofs the first predict sees and -- frozen DOFs are zeroed by it -- at the ones every later
        // predict (and every Jacobian block) sees
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (!(targ[i] == dofs[i])) { targ[i] = dofs[i]; t_sincos(dofs[i], ts[i], tcs[i]); }
        ProbeTrig<T> t0, t1;
        t0.s1 = ts[0]; t0.c1 = tcs[0]; t0.s2 = ts[1]; t0.c2 = tcs[1]; t0.s3 = ts[2]; t0.c3 = tcs[2];
        t0.sp = t0.cp = T(0);
        t1 = t0;
        if (zeroed[0]) { t1.s1 = T(0); t1.c1 = T(1); }
        if (zeroed[1]) { t1.s2 = T(0); t1.c2 = T(1); }
        if (zeroed[2]) { t1.s3 = T(0); t1.c3 = T(1); }
        auto probe_p = [&](const ProbeTrig<T>& t, const T* d, T* pB) {
            T z6[3], e5[3], e6[3];
            z6[0] = -t.s1 * t.c2; z6[1] = -t.c1 * t.c2; z6[2] = t.s2;
            E5(t.s3, t.c3, e5); E6(t.s3, t.c3, e6);
            const T lq = prm.L - d[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) pB[i] = lq * z6[i] + d[4] * e5[i] + d[5] * e6[i];
        };
        T pB0[3], pB1[3];
        probe_p(t0, dofs, pB0);
        if (n > 0) {                                    // (an update-only epoch leaves the extrinsics alone)
#pragma unroll
            for (int i = 0; i < 6; ++i) if (zeroed[i]) dofs[i] = T(0);
        }
        probe_p(t1, dofs, pB1);
        mb[MB_T0 + 0] = t0.s1; mb[MB_T0 + 1] = t0.c1; mb[MB_T0 + 2] = t0.s2; mb[MB_T0 + 3] = t0.c2;
        mb[MB_T1 + 0] = t1.s1; mb[MB_T1 + 1] = t1.c1; mb[MB_T1 + 2] = t1.s2; mb[MB_T1 + 3] = t1.c2; mb[MB_T1 + 4] = t1.s3; mb[MB_T1 + 5] = t1.c3;
#pragma unroll
        for (int i = 0; i < 6; ++i) mb[MB_DOFS + i] = dofs[i];
        if (lane == 0) publish(flags + F_RA_CONST, e + 1);
        // trajectory log: the snapshot the update before belongs to gets the corrected state (Filter.update overwrites the
        // last row, Filter.py:393)
        if (A.xlog_mode && live && b < A.n_log && e > 0 && (A.xlog_mode == 2 || g > 0)) {
            T* xl = reinterpret_cast<T*>(A.xlog) + (int64_t)(A.xlog_mode == 2 ? e - 1 : g - 1) * 26 * A.n_log + b;
#pragma unroll
            for (int i = 0; i < 3; ++i) { xl[(0 + i) * A.n_log] = p[i]; xl[(3 + i) * A.n_log] = v[i]; xl[(19 + i) * A.n_log] = pc[i]; }
#pragma unroll
            for (int i = 0; i < 4; ++i) xl[(6 + i) * A.n_log] = q[i];
#pragma unroll
            for (int i = 0; i < 6; ++i) xl[(10 + i) * A.n_log] = dofs[i];
        }
        for (int k = 0; k < n; ++k, ++g) {
            T dt2, om2[3], acc2[3];                     // the next sample is on its way while this predict is computed
            load_sample(a, g + 1, g_end, b, dt2, om2, acc2);
            // what the Jacobian blocks of this predict are linearised with: the buffers as they are BEFORE it
            // (Filter.py:220-221,254-255,325-327) -- published first, the J wavefronts start on them at once
            if (g >= L::DA) {
                EP_WAIT_BEGIN()
                wait_ge(flags + F_JA, g - L::DA + 1);
                wait_ge(flags + F_JB, g - L::DA + 1);
                EP_WAIT_END(1)
            }
            {
                T* ax = aux + ((g % L::DA) * NI + lane) * AW;
#pragma unroll
                for (int i = 0; i < 4; ++i) ax[AX_QO + i] = qo[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) { ax[AX_OM + i] = om_o[i]; ax[AX_ACC + i] = acc_o[i]; }
                ax[AX_DT] = dt;
            }
            if (lane == 0) publish(flags + F_RA, g + 1);
            T R[9], om_avg[3], w[3], RS[9], Rn[9], a1[3], a2[3], acc_avg[3], cr[3], Rcr[3];
            quat_to_mat(q, R);
#pragma unroll
            for (int i = 0; i < 3; ++i) { om_avg[i] = (om_o[i] + om_n[i]) * T(0.5); w[i] = dt * om_avg[i]; }
            mat3_skew(R, w, RS);
#pragma unroll
            for (int i = 0; i < 9; ++i) Rn[i] = R[i] + RS[i];                       // equations.py:46
            mat3_vec(R, acc_o, a1);
            mat3_vec(Rn, acc_n, a2);
#pragma unroll
            for (int i = 0; i < 3; ++i) acc_avg[i] = (a1[i] + a2[i]) * T(0.5);      // equations.py:47
            cross3(om_avg, k == 0 ? pB0 : pB1, cr);                                 // BB_v_CB == 0
            mat3_vec(R, cr, Rcr);
            advance_p_v_pc(dt, acc_avg, Rcr, p, v, pc);                               // equations.py:76-77,83-85
#pragma unroll
            for (int i = 0; i < 3; ++i) { om_o[i] = om_n[i]; acc_o[i] = acc_n[i]; }  // Filter.py:225-226
            mat_to_quat(Rn, q);                                                     // state.py:67
#pragma unroll
            for (int i = 0; i < 4; ++i) qo[i] = q[i];                               // = R_WB_old of the next predict (Filter.py:227)
            if (A.xlog_mode == 1 && live && b < A.n_log) {                          // Filter.py:230: the propagated state
                T* xl = reinterpret_cast<T*>(A.xlog) + (int64_t)g * 26 * A.n_log + b;
#pragma unroll
                for (int i = 0; i < 3; ++i) { xl[(0 + i) * A.n_log] = p[i]; xl[(3 + i) * A.n_log] = v[i]; xl[(19 + i) * A.n_log] = pc[i]; }
#pragma unroll
                for (int i = 0; i < 4; ++i) xl[(6 + i) * A.n_log] = q[i];
#pragma unroll
                for (int i = 0; i < 6; ++i) xl[(10 + i) * A.n_log] = dofs[i];
            }
            dt = dt2;
#pragma unroll
            for (int i = 0; i < 3; ++i) { om_n[i] = om2[i]; acc_n[i] = acc2[i]; }
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) mb[MB_PC + i] = gpos[i] - pc[i];                // Filter.py:369 (RB adds it to the residual)
        if (lane == 0) publish(flags + F_RA_EP, e + 1);
    }
    if (live) {
        T* xo = reinterpret_cast<T*>(a.xs);
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            xo[(0 + i) * B + b] = p[i]; xo[(3 + i) * B + b] = v[i]; xo[(19 + i) * B + b] = pc[i];
            xo[(26 + i) * B + b] = om_o[i]; xo[(29 + i) * B + b] = acc_o[i];
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { xo[(6 + i) * B + b] = q[i]; xo[(32 + i) * B + b] = qo[i]; }
#pragma unroll
        for (int i = 0; i < 6; ++i) xo[(10 + i) * B + b] = dofs[i];
    }
}

// ---- RB: camera attitude, notch states, residual (eskf_nominal_kernel, role 1) ----------------------------------
template <typename T, typename L>
__device__ __forceinline__ void role_rb(const EpochsArgs& A, const DevParams<T>& prm, T* lds, const int lane, const int64_t b0, const int ni,
                                        const int nca, unsigned long long* wait_cycles) {
    constexpr int NI = L::NI;
    if (lane >= NI) return;
    const StepArgs& a = A.a;
    T* aux = lds + L::O_AUX;
    T* mb = lds + L::O_MBOX + lane * MW;
    int* flags = reinterpret_cast<int*>(lds + L::O_FLAGS);
    const bool live = lane < ni;
    const int64_t b = b0 + (live ? lane : ni - 1);
    const int64_t B = a.B;
    const T* xi = reinterpret_cast<const T*>(a.xs_src);
    auto X = [&](int w) { return xi[(int64_t)w * B + b]; };
    bool zeroed[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) zeroed[i] = !prm.hold_frozen && ((prm.frozen_mask >> i) & 1);
    T qc[4], notch[3], om_o[3], q3;
#pragma unroll
    for (int i = 0; i < 3; ++i) { notch[i] = X(16 + i); om_o[i] = X(26 + i); }
#pragma unroll
    for (int i = 0; i < 4; ++i) qc[i] = X(22 + i);
    q3 = X(12);
    T ph_arg = T(NAN), sp = T(0), cp = T(1);            // sin / cos of the phi they were last evaluated at
    int g = 0;
    for (int e = 0; e < A.n_epochs; ++e) {
        const int n = n_pred_of(A, e);
        const int g_end = g + n;
        T qraw[4], ang_notch;
        {
            const int64_t mo = (int64_t)e * (a.meas_stride ? B : 1) + (a.meas_stride ? b : 0);
            const T* gq = reinterpret_cast<const T*>(a.quat) + mo * 4;
#pragma unroll
            for (int i = 0; i < 4; ++i) qraw[i] = gq[i];
            ang_notch = reinterpret_cast<const T*>(a.notch)[mo];
        }
        T dt, om_n[3], acc_n[3];
        load_sample(a, g, g_end, b, dt, om_n, acc_n);
        Pending<T> pd;
        T qm[4];                                                                        // Filter.py:364-365
        {   // (depends on the measurement only: evaluated before the wait for the update)
            T nq[4] = {T(0), T(0), T(0), T(1)};
            t_sincos(T(0.5) * ang_notch, nq[2], nq[3]);          // Quaternion(val=[0,0,ang], euler="xyz")
            quat_mul(nq, qraw, qm);
            quat_nrm(qm);
            qm[0] = -qm[0]; qm[1] = -qm[1]; qm[2] = -qm[2];      // conjugate
        }
        EP_WAIT_BEGIN()
        pd = pending_of<T, 6, 3, 11, 1, 15, 3, 21, 3>(a, e, b, mb, flags, nca, lane);
        EP_WAIT_END(0)
        if (pd.inject) {
            T e69[3], e1518[3], e2124[3], e11 = pd.src[11];
#pragma unroll
            for (int i = 0; i < 3; ++i) { e69[i] = pd.src[6 + i]; e1518[i] = pd.src[15 + i]; e2124[i] = pd.src[21 + i]; }
            T dqc[4], qcn[4];
            quat_about_axis(norm3(e2124), (prm.quirks & QK_Q3) ? e69 : e2124, dqc);   // sic (state.py:124) unless Q3 is off
            quat_mul(qc, dqc, qcn);
            quat_nrm(qcn);
#pragma unroll
            for (int i = 0; i < 4; ++i) qc[i] = qcn[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) notch[i] += e1518[i];
            q3 += e11;
        }
        if (A.xlog_mode && live && b < A.n_log && e > 0 && (A.xlog_mode == 2 || g > 0)) {      // (see role_ra)
            T* xl = reinterpret_cast<T*>(A.xlog) + (int64_t)(A.xlog_mode == 2 ? e - 1 : g - 1) * 26 * A.n_log + b;
#pragma unroll
            for (int i = 0; i < 3; ++i) xl[(16 + i) * A.n_log] = notch[i];
#pragma unroll
            for (int i = 0; i < 4; ++i) xl[(22 + i) * A.n_log] = qc[i];
        }
        EP_WAIT_BEGIN()
        wait_ge(flags + F_RA_CONST, e + 1);
        EP_WAIT_END(2)
        T s1 = mb[MB_T0 + 0], c1 = mb[MB_T0 + 1], s2 = mb[MB_T0 + 2], c2 = mb[MB_T0 + 3];
        for (int k = 0; k < n; ++k, ++g) {
            T dt2, om2[3], acc2[3];
            load_sample(a, g + 1, g_end, b, dt2, om2, acc2);
            if (k == 1) {                                   // (from the second predict on the frozen DOFs are zero)
                if (zeroed[0]) { s1 = T(0); c1 = T(1); }
                if (zeroed[1]) { s2 = T(0); c2 = T(1); }
                if (zeroed[2]) q3 = T(0);
            }
            const T phi = q3 - notch[0];
            if (!(phi == ph_arg)) { ph_arg = phi; t_sincos(phi, sp, cp); }
            ProbeTrig<T> t;
            t.s1 = s1; t.c1 = c1; t.s2 = s2; t.c2 = c2; t.s3 = t.c3 = T(0); t.sp = sp; t.cp = cp;
            T RBC[9], z6[3];
            {
                T aa[3], gg[3];
                z6[0] = -t.s1 * t.c2; z6[1] = -t.c1 * t.c2; z6[2] = t.s2;
                E5(t.sp, t.cp, aa); E6(t.sp, t.cp, gg);
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    RBC[3 * i + 0] = -aa[i];
                    RBC[3 * i + 1] = prm.ct * gg[i] + prm.st * z6[i];
                    RBC[3 * i + 2] = -prm.st * gg[i] + prm.ct * z6[i];
                }
            }
            T u[3], C_om[3], wc[3], Rc[9], RcS[9], Rcn[9];
#pragma unroll
            for (int i = 0; i < 3; ++i) u[i] = ((prm.quirks & QK_Q9) ? om_o[i] : (om_o[i] + om_n[i]) * T(0.5)) + z6[i] * notch[1];   // (Q9)
            mat3T_vec(RBC, u, C_om);                                                    // equations.py:50
#pragma unroll
            for (int i = 0; i < 3; ++i) wc[i] = dt * C_om[i];
            quat_to_mat(qc, Rc);
            mat3_skew(Rc, wc, RcS);
#pragma unroll
            for (int i = 0; i < 9; ++i) Rcn[i] = Rc[i] + RcS[i];                        // equations.py:86
            mat_to_quat(Rcn, qc);                                                       // state.py:72
            notch[0] += dt * notch[1];                                                  // equations.py:80-82
            notch[1] += dt * notch[2];
            // what the Jacobian blocks of this predict are evaluated at: the dofs after it (frozen ones zeroed), the notch after it
            const T phi_n = (zeroed[2] ? T(0) : q3) - notch[0];
            if (!(phi_n == ph_arg)) { ph_arg = phi_n; t_sincos(phi_n, sp, cp); }
            if (g >= L::DA) {
                EP_WAIT_BEGIN()
                wait_ge(flags + F_JB, g - L::DA + 1);
                EP_WAIT_END(1)
            }
            {
                T* ax = aux + ((g % L::DA) * NI + lane) * AW;
                ax[AX_N0] = notch[0]; ax[AX_N1] = notch[1]; ax[AX_SP] = sp; ax[AX_CP] = cp;
            }
            if (lane == 0) publish(flags + F_RB, g + 1);
            if (A.xlog_mode == 1 && live && b < A.n_log) {
                T* xl = reinterpret_cast<T*>(A.xlog) + (int64_t)g * 26 * A.n_log + b;
#pragma unroll
                for (int i = 0; i < 3; ++i) xl[(16 + i) * A.n_log] = notch[i];
#pragma unroll
                for (int i = 0; i < 4; ++i) xl[(22 + i) * A.n_log] = qc[i];
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) { om_o[i] = om_n[i]; om_n[i] = om2[i]; }
            dt = dt2;
        }
        if (n > 0 && zeroed[2]) q3 = T(0);                  // (what the state's q3 is after a predict)
        // residual (Filter.py:369-375); r_p comes from RA
        {
            T qe[4];
            quat_mul(qm, qc, qe);
            quat_nrm(qe);
            T nv = t_sqrt(qe[0] * qe[0] + qe[1] * qe[1] + qe[2] * qe[2]);
            T flag = T(0);
            if (nv > T(1)) { nv = T(1); flag = T(1); }
            const T ang = residual_angle(nv, prm.quirks);             // half-angle residual (Quaternion.py:138-147) unless Q6 is off
            const T f = (nv == T(0)) ? T(0) : ang / nv;
            EP_WAIT_BEGIN()
            wait_ge(flags + F_RA_EP, e + 1);
            EP_WAIT_END(3)
            T r[7];
#pragma unroll
            for (int i = 0; i < 3; ++i) { r[i] = mb[MB_PC + i]; r[3 + i] = f * qe[i]; }
            r[6] = ang_notch - notch[0];
            // a NaN / inf measurement or state propagates exactly as it does in the reference (NumPy arithmetic, no
            // check); the instance is flagged, its neighbours are not affected
            if (!t_finite(r[0] + r[1] + r[2] + r[3] + r[4] + r[5] + r[6])) flag = T(1);
#pragma unroll
            for (int i = 0; i < 7; ++i) mb[MB_RES + i] = r[i];
            mb[MB_RES + 7] = flag;                                // asin argument was clamped, or non-finite residual
        }
        if (lane == 0) publish(flags + F_RB_EP, e + 1);
    }
    if (live) {
        T* xo = reinterpret_cast<T*>(a.xs);
#pragma unroll
        for (int i = 0; i < 3; ++i) xo[(16 + i) * B + b] = notch[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) xo[(22 + i) * B + b] = qc[i];
    }
}

// ---- JA / JB: the Jacobian blocks of up to PPP predicts at once (eskf_jac_kernel's arithmetic) --------------------
template <typename T, typename L, int WHICH>
__device__ __forceinline__ void role_j(const EpochsArgs& A, const DevParams<T>& prm, T* lds, const int lane, const int nca,
                                       unsigned long long* wait_cycles) {
    constexpr int NI = L::NI, NIP = L::NIP, PPP = L::PPP, D = L::D, DA = L::DA;
    const T* aux = lds + L::O_AUX;
    T* ring = lds + L::O_RING;
    int* flags = reinterpret_cast<int*>(lds + L::O_FLAGS);
    const int ps = lane / NIP, i = lane % NIP;
    const bool lane_ok = i < NI && ps < PPP;
    const T* mb = lds + L::O_MBOX + (i < NI ? i : 0) * MW;
    const T sig_om[3] = {prm.sig_om[0], prm.sig_om[1], prm.sig_om[2]};
    int g0 = 0;
    while (g0 < A.n_total) {
        // How many predicts this pass takes: what the recurrences have published, what the ring has room for -- and, while
        // no covariance wavefront is about to run out of records, not fewer than a full pass (PPP predicts, one per group of
        // NIP lanes: a pass costs the same instructions whether one group works or all of them, and with the ring full the
        // wavefront would otherwise wake up for every single slot that becomes free).
        int n;
        {
            const int n_target = A.n_total - g0 < PPP ? A.n_total - g0 : PPP;
            EP_WAIT_BEGIN()
            for (;;) {
                int have = peek(flags + F_RA);
                if (WHICH) { const int hb = peek(flags + F_RB); have = hb < have ? hb : have; }
                n = have - g0;
                n = n < n_target ? n : n_target;
                // record g lies in slot g % D: free once every covariance wavefront has taken record g - D
                while (n > 0 && !all_ge(flags + F_LOADED, nca, g0 + n - D, lane)) --n;
                if (n >= n_target) break;
                // (somebody has at most one record left in front of it -- g0 - 1 is the last one published: do not let it wait)
                if (n > 0 && any_ge(flags + F_LOADED, nca, g0 - 1, lane)) break;
                __builtin_amdgcn_s_sleep(1);
            }
            EP_WAIT_END(0)
            asm volatile("" ::: "memory");
        }
        if (lane_ok && ps < n) {
            const int g = g0 + ps;
            const T* ax = aux + ((g % DA) * NI + i) * AW;
            Jac<T>& J = *reinterpret_cast<Jac<T>*>(ring + ((g % D) * NI + i) * RW);
            const T dt = ax[AX_DT];
            T om_o[3], om_t[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { om_o[k] = ax[AX_OM + k]; om_t[k] = om_o[k] - sig_om[k]; }
            ProbeTrig<T> t;
            t.s1 = mb[MB_T1 + 0]; t.c1 = mb[MB_T1 + 1]; t.s2 = mb[MB_T1 + 2]; t.c2 = mb[MB_T1 + 3];
            t.s3 = mb[MB_T1 + 4]; t.c3 = mb[MB_T1 + 5];
            if constexpr (WHICH == 0) {
                t.sp = t.cp = T(0);
                T acc_o[3], qo[4], dofs[6], Ro[9];
#pragma unroll
                for (int k = 0; k < 3; ++k) acc_o[k] = ax[AX_ACC + k];
#pragma unroll
                for (int k = 0; k < 4; ++k) qo[k] = ax[AX_QO + k];
#pragma unroll
                for (int k = 0; k < 6; ++k) dofs[k] = mb[MB_DOFS + k];
                jac_blocks_old_buffers(qo, acc_o, om_o, dt, J, Ro);
                jac_blocks_position(prm, t, dofs, om_t, Ro, dt, J);
            } else {
                t.sp = ax[AX_SP]; t.cp = ax[AX_CP];
                jac_blocks_attitude(prm, t, ax[AX_N1], om_t, om_o, dt, J);
            }
        }
#ifdef ESKF_X_EPDUMP     // diagnostic build: the records as the covariance wavefronts will see them, [predict][instance][67]
        if (A.stamps && lane_ok && ps < n && i < NI) {
            const int g = g0 + ps;
            const int64_t bb = (int64_t)blockIdx.x * NI + i;
            if (bb < A.a.B) {
                const T* rec = ring + ((g % D) * NI + i) * RW;
                T* out = reinterpret_cast<T*>(A.stamps) + ((int64_t)g * A.a.B + bb) * RW;
                for (int w = (WHICH ? J_DM - J_B0 : 0); w < (WHICH ? J_DT - J_B0 : J_DM - J_B0); ++w) out[w] = rec[w];
                if (!WHICH) out[J_DT - J_B0] = rec[J_DT - J_B0];
            }
        }
#endif
        g0 += n;
        if (lane == 0) publish(flags + (WHICH ? F_JB : F_JA), g0);
    }
}

// ---- C: the covariance of PP pairs of instances, rows in registers for the whole run -----------------------------
// (the 32-lanes-per-instance / DPP / SLIM path of cov_rows_body, cut at its phase boundaries: load | predict | update +
//  reset | store -- same statements in the same order.  PP = 2: the wavefront does every phase for its first pair, then
//  for its second -- two independent instruction streams between the same synchronisation points)
template <typename T, typename L>
__device__ __forceinline__ void role_c(const EpochsArgs& A, const DevParams<T>& prm, T* lds, const int cw, const int lane_in, const int64_t b0,
                                       const int ni, unsigned long long* wait_cycles) {
    using namespace rowk;
    constexpr int NI = L::NI, D = L::D, PP = L::PP;
    constexpr int PW = 16 / (int)sizeof(T);         // words per 16-byte piece of the packed record
    constexpr int NPC = 300 / PW;                   // pieces per record
    constexpr int NV = (NPC + 31) / 32;             // pieces per lane
    using TP = typename Vec16<T>::type;
    const StepArgs& a = A.a;
    int lane = lane_in;
    asm volatile("" : "+v"(lane));
    const int grp = lane >> 5, lig = lane & 31;
    const bool active = lig < 24;
    int row = active ? lig : 23;                    // (idle lanes shadow row 23 and write nothing)
    const T* ring = lds + L::O_RING;
    int* flags = reinterpret_cast<int*>(lds + L::O_FLAGS);
    T* sB[PP];                                      // the pair's exchange images
    T* mb[PP];
    bool valid[PP];
    int ilc[PP];
    int64_t inst[PP];
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        const int il = (cw * PP + pp) * 2 + grp;    // the instance's number in the block
        valid[pp] = il < ni;
        ilc[pp] = valid[pp] ? il : ni - 1;          // (a ragged block's last lanes recompute its last instance, write nothing)
        inst[pp] = b0 + ilc[pp];
        sB[pp] = lds + L::O_IMG + il * IMG;
        mb[pp] = lds + L::O_MBOX + ilc[pp] * MW;
    }
    asm volatile("" : "+v"(row));
    constexpr int sel[7] = {18, 19, 20, 21, 22, 23, 15};
    T p[PP][24];                                    // row `row` of P, per pair
    // ---- load: packed record -> LDS image in its own order -> the lane's row (cov_rows_body, ESKF_ROWS_VIA_LDS)
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        typedef T row_piece_t __attribute__((ext_vector_type(PW)));
        row_piece_t pv_[NV];
        const char* gPb_ = reinterpret_cast<const char*>(reinterpret_cast<const T*>(a.P_src) + inst[pp] * 300);
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int e_ = lig + k * 32;
            pv_[k] = *reinterpret_cast<const row_piece_t*>(gPb_ + (size_t)(16u * (unsigned)(e_ < NPC ? e_ : NPC - 1)));
        }
#pragma unroll
        for (int k = 0; k < NV; ++k) {
            const int e_ = lig + k * 32;
            *reinterpret_cast<row_piece_t*>(sB[pp] + PW * (e_ < NPC ? e_ : NPC - 1)) = pv_[k];
        }
        wave_sync();
        const unsigned rw_ = (unsigned)row;
        const unsigned lb_ = (unsigned)(size_t)(__attribute__((address_space(3))) T*)sB[pp];
        constexpr unsigned ESZ = (unsigned)sizeof(T);
        const unsigned va_ = lb_ + ESZ * (rw_ < 12u ? 25u * rw_ : 26u * (23u - rw_) + 1u), vb_ = lb_ + ESZ * rw_;
        gather_row_masked(va_, vb_, p[pp]);
        wave_sync();
    }
    int g = 0;
    for (int e = 0; e < A.n_epochs; ++e) {
        const int n = n_pred_of(A, e);
        // ================================ predicts =======================================
        // The record of predict g and the two counters that say whether it is complete come in ONE LDS round trip (the
        // counters are read first: LDS executes a wavefront's reads in order, so a record read behind counters that say
        // "complete" is the complete record), requested while the predict before is still computing its second Fx
        // application; only a record that was not ready then is asked for again, polling.
        T jn[PP][3];
#pragma unroll
        for (int pp = 0; pp < PP; ++pp) jn[pp][0] = jn[pp][1] = jn[pp][2] = T(0);
        int fa = 0, fb = 0;
        auto fetch_record = [&](int gg) {
            typedef const volatile __attribute__((address_space(3))) T lds_word_t;
            typedef const volatile __attribute__((address_space(3))) int lds_flag_t;
            const int w0 = lig >= J_B0 ? lig - J_B0 : 0;
            const int w2 = lig + 64 - J_B0 < RW - 1 ? lig + 64 - J_B0 : RW - 1;
            fa = *(lds_flag_t*)(flags + F_JA);
            fb = *(lds_flag_t*)(flags + F_JB);
            // lane l holds words l, 32 + l, 64 + l of the 75-word record of the two-kernel path, whose words 0..7 are the
            // residual; the ring keeps the 67 block words
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) {
                const T* rec = ring + ((gg % D) * NI + ilc[pp]) * RW;
                jn[pp][0] = *(lds_word_t*)(rec + w0);
                jn[pp][1] = *(lds_word_t*)(rec + (32 + lig - J_B0));
                jn[pp][2] = *(lds_word_t*)(rec + w2);
            }
        };
        auto record_ready = [&](int gg) {
            return __builtin_amdgcn_readfirstlane(fa) > gg && __builtin_amdgcn_readfirstlane(fb) > gg;
        };
        if (n > 0) fetch_record(g);
        for (int k = 0; k < n; ++k, ++g) {
            EP_WAIT_BEGIN()
            while (!record_ready(g)) {
                __builtin_amdgcn_s_sleep(1);
                fetch_record(g);
            }
            EP_WAIT_END(0)
            T j0[PP][3];
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) {
                j0[pp][0] = jn[pp][0]; j0[pp][1] = jn[pp][1]; j0[pp][2] = jn[pp][2];
                asm volatile("" : "+v"(j0[pp][0]), "+v"(j0[pp][1]), "+v"(j0[pp][2]) : : "memory");
            }
            if (lane == 0) publish(flags + F_LOADED + cw, g + 1);
            JRegs<T> Jr[PP];
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) {
#pragma unroll
                for (int q = 0; q < 3; ++q) spread_rows(j0[pp][q], Jr[pp].X[q], Jr[pp].Y[q]);
                fx_apply_store_dpp(Jr[pp], p[pp], sB[pp] + row * PS, active);  // column `row` of Fx P = row `row` of W = P Fx^T
            }
            wave_sync();
            if (k + 1 < n) fetch_record(g + 1);
#pragma unroll
            for (int pp = 0; pp < PP; ++pp) {
                fx_apply_col_dpp(Jr[pp], sB[pp] + row, p[pp]);                 // column `row` of Fx P Fx^T = row `row` of P'
                noise_diag_exec(prm, p[pp]);
            }
            wave_sync();
        }
        // ================================ update ===========================================
        // (cov_rows_body: the 7 measurements one after the other; sv is the pivot d_m of S = L D L^T; a zero or non-finite
        //  pivot is the reference's LinAlgError, Filter.py:356-361)
        EP_WAIT_BEGIN()
        wait_ge(flags + F_RB_EP, e + 1);
        EP_WAIT_END(1)
        const bool last = e + 1 == A.n_epochs;
        T eev[PP];                                  // component `row` of the error state, per pair
        bool updv[PP];
        static_for<0, PP>([&](auto PPc) {
            constexpr int pp = decltype(PPc)::value;
            T (&pr)[24] = p[pp];
            T* const sBp = sB[pp];
            T Xres = mb[pp][MB_RES + (lig & 7)];    // lanes 0..7 of each 16-lane row: the residual and its flag
            asm("s_nop 1" : "+v"(Xres));
            T ee = T(0);
            int status = 0;
            bool fail = false;
            {
                status = (dpp_bcast<7>(Xres) != T(0)) ? DVI_STATUS_NONFINITE : 0;
                T E[7];
                int napplied = 0;
                unsigned long long okm[7] = {0, 0, 0, 0, 0, 0, 0};
                unsigned long long failm = 0;
                constexpr int CS = 32;
                static_for<0, 7>([&](auto M) {
                    constexpr int m = decltype(M)::value;
                    constexpr int sm = sel[m];
                    const T pm = pr[sm];
                    lds_store1(sBp + m * CS + lig, pm);
                    T X, Y;                     // column sel_m of the current P: X rows 0..15, Y rows 16..23
                    spread_rows(pm, X, Y);
                    const T csel = dpp_bcast<(sm & 15)>(sm < 16 ? X : Y);
                    const T sv = csel + prm.r[m];
                    failm |= unusable_pivot_mask(sv);
                    okm[m] = ~failm;
                    if (!__builtin_amdgcn_inverse_ballot_w64(failm)) {
                        const T gg = fast_rcp(sv);
                        const T rm = dpp_bcast<m>(Xres);
                        const T tau = (m == 0) ? rm * gg : (rm - E[m]) * gg;
                        const T nkk = pm * -gg;
                        if constexpr (m == 0) ee = pm * tau;
                        else ee += pm * tau;
                        static_for<m + 1, 7>([&](auto M2) {
                            constexpr int s2 = sel[decltype(M2)::value];
                            dpp_fmac<(s2 & 15), m == 0>(E[decltype(M2)::value], s2 < 16 ? X : Y, tau);
                        });
                        static_for<12, 24>([&](auto J) {
                            constexpr int j = decltype(J)::value;
                            dpp_fmac<(j & 15)>(pr[j], j < 16 ? X : Y, nkk);
                        });
                        static_for<0, 12>([&](auto J) {
                            constexpr int j = decltype(J)::value;
                            dpp_fmac<(j & 15)>(pr[j], X, nkk);
                        });
                    }
                    pin_row(pr);
                    asm volatile("" : "+v"(ee), "+v"(E[0]), "+v"(E[1]), "+v"(E[2]), "+v"(E[3]), "+v"(E[4]), "+v"(E[5]), "+v"(E[6]));
                });
                fail = __builtin_amdgcn_inverse_ballot_w64(failm);
                if (__builtin_expect(fail, 0)) wave_sync();
                if (__builtin_expect(fail, 0)) {
                    status |= DVI_STATUS_SINGULAR;
#pragma unroll
                    for (int m = 0; m < 7; ++m) napplied += (int)((okm[m] >> lane) & 1ull);
#pragma unroll
                    for (int m = 6; m >= 0; --m) {
                        if (m < napplied) {
                            const T* cb = sBp + m * CS;
                            const T gg = fast_rcp(cb[sel[m]] + prm.r[m]);
                            const T kk = cb[row] * gg;
#pragma unroll
                            for (int j = 0; j < 24; ++j) pr[j] += kk * cb[j];
                            pin_row(pr);
                        }
                    }
                }
            }
            const bool upd = !fail;
            if (row >= 9 && row < 15 && ((prm.frozen_mask >> (row - 9)) & 1)) ee = T(0);   // Filter.py:377-379
            // The error state and the status word are final here: they go to the recurrences through the mailbox BEFORE the
            // reset of the covariance, which then runs next to the injection (State.__add__) on the producer side; after the
            // run's last update they go to memory, where the next call (or dvi_eskf_flush) finds them -- what the epochs
            // enqueued one by one leave behind
            if (valid[pp]) {
                if (lig < 24) mb[pp][MB_E + lig] = upd ? ee : T(0);
                if (lig == 24) mb[pp][MB_ST] = status_as_word(status | DVI_STATUS_PENDING, T(0));
                if (last && upd && active) reinterpret_cast<T*>(a.e)[inst[pp] * 24 + row] = ee;
                if (lig == 0) {
                    if (last) a.status_ws[inst[pp]] = status | DVI_STATUS_PENDING;
                    if (a.status) a.status[(int64_t)e * a.B + inst[pp]] = status;
                }
            }
            eev[pp] = ee;
            updv[pp] = upd;
        });
        if (lane == 0) publish(flags + F_CEPOCH + cw, e + 1);
        static_for<0, PP>([&](auto PPc) {
            constexpr int pp = decltype(PPc)::value;
            T (&pr)[24] = p[pp];
            T* const sBp = sB[pp];
            const T ee = eev[pp];
            const bool upd = updv[pp];
            T Gt[3], Gc[3];
            {
                T Xe, Ye;
                spread_rows(ee, Xe, Ye);
                Gt[0] = T(0.5) * dpp_bcast<6>(Xe); Gt[1] = T(0.5) * dpp_bcast<7>(Xe); Gt[2] = T(0.5) * dpp_bcast<8>(Xe);
                Gc[0] = T(0.5) * dpp_bcast<5>(Ye); Gc[1] = T(0.5) * dpp_bcast<6>(Ye); Gc[2] = T(0.5) * dpp_bcast<7>(Ye);
            }
            T* sY = sBp;    // [24][XS]: columns 6:9 -> words 0:3, columns 21:24 -> words 4:7
            if (upd) {
                // ---- reset G P G^T (Filter.py:387-390), G = I - [theta/2]x on 6:9 and 21:24: column part, in registers
                const T a0 = pr[6], a1 = pr[7], a2 = pr[8];
                pr[6] = a0 + Gt[2] * a1 - Gt[1] * a2;
                pr[7] = a1 - Gt[2] * a0 + Gt[0] * a2;
                pr[8] = a2 + Gt[1] * a0 - Gt[0] * a1;
                const T c0 = pr[21], c1 = pr[22], c2 = pr[23];
                pr[21] = c0 + Gc[2] * c1 - Gc[1] * c2;
                pr[22] = c1 - Gc[2] * c0 + Gc[0] * c2;
                pr[23] = c2 + Gc[1] * c0 - Gc[0] * c1;
                if (active) {
#pragma unroll
                    for (int k = 0; k < 3; ++k) {
                        sY[row * XS + k] = pr[6 + k];
                        sY[row * XS + 4 + k] = pr[21 + k];
                    }
                }
            }
            wave_sync();
            // row part: only rows 6:9 and 21:24 change (see cov_rows_body)
            const bool lo = row >= 6 && row < 9, hi = row >= 21;
            if (upd && (lo || hi)) {
                const int tb = lo ? 6 : 21;
                const int t = row - tb;
                const int sw = lo ? t : 4 + t;
                const T g0 = lo ? Gt[0] : Gc[0], g1 = lo ? Gt[1] : Gc[1], g2 = lo ? Gt[2] : Gc[2];
                const T k0 = (t == 0) ? T(1) : (t == 1) ? -g2 : g1;
                const T k1 = (t == 0) ? g2 : (t == 1) ? T(1) : -g0;
                const T k2 = (t == 0) ? -g1 : (t == 1) ? g0 : T(1);
                const T* y0 = sY + tb * XS;
#pragma unroll
                for (int j = 0; j < 24; ++j) {
                    const bool jl = j >= 6 && j < 9, jh = j >= 21;
                    if (jl || jh) {
                        const int sj = jl ? j - 6 : 4 + j - 21;
                        pr[j] = k0 * lds_load1(y0 + sj) + k1 * lds_load1(y0 + XS + sj) + k2 * lds_load1(y0 + 2 * XS + sj);
                    } else {
                        pr[j] = lds_load1(sY + j * XS + sw);
                    }
                }
            }
            wave_sync();
            // The covariance is symmetric-packed wherever it is at rest: an epoch enqueued on its own
            // stores the lower triangle and the next one reads P[i][j] = packed[max(i, j)][min(i, j)].  The rows a lane
            // computes are not symmetric to the last bit, so the same is done here between two epochs -- through the image,
            // the store phase's packed order and the load phase's gather -- or a run would depend on where it is cut into calls.
            if (!last) {
                const int st_ = row < 12 ? 25 * row : 26 * (23 - row) + 1;
                T* rp_ = sBp + st_;
                store_row_masked((unsigned)(size_t)(__attribute__((address_space(3))) T*)rp_, pr);
                wave_sync();
                const unsigned rw_ = (unsigned)row;
                const unsigned lb_ = (unsigned)(size_t)(__attribute__((address_space(3))) T*)sBp;
                constexpr unsigned ESZ = (unsigned)sizeof(T);
                const unsigned va_ = lb_ + ESZ * (rw_ < 12u ? 25u * rw_ : 26u * (23u - rw_) + 1u), vb_ = lb_ + ESZ * rw_;
                gather_row_masked(va_, vb_, pr);
                wave_sync();
            }
        });
    }
    // ---- store: rows -> LDS in the order of the packed record -> the record as a straight copy ------------------------
#pragma unroll
    for (int pp = 0; pp < PP; ++pp) {
        const int st_ = row < 12 ? 25 * row : 26 * (23 - row) + 1;
        T* rp_ = sB[pp] + st_;
        store_row_masked((unsigned)(size_t)(__attribute__((address_space(3))) T*)rp_, p[pp]);
        wave_sync();
        if (valid[pp]) {
            char* gPb_ = reinterpret_cast<char*>(reinterpret_cast<T*>(a.P) + inst[pp] * 300);
            TP v[NV];
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int e_ = lig + k * 32;
                typedef T piece_vec_t __attribute__((ext_vector_type(PW)));
                typedef const volatile __attribute__((address_space(3))) piece_vec_t lds_piece_t;
                const piece_vec_t t_ = *(lds_piece_t*)(sB[pp] + PW * (e_ < NPC ? e_ : NPC - 1));
                __builtin_memcpy(&v[k], &t_, sizeof(TP));
            }
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int e_ = lig + k * 32;
                if (e_ < NPC) *reinterpret_cast<TP*>(gPb_ + (size_t)(16 * e_)) = v[k];
            }
        }
    }
}

}  // namespace ep

// Which wavefront of the block does what.  A block's wavefronts go to the CU's four SIMDs in turn.  With two covariance
// wavefronts per block (the small batches, where a covariance wavefront's latency is all there is) the block has eight
// wavefronts of which two leave at once, so that each covariance wavefront has a SIMD to itself and the four producers
// share the other two: 0 C0 | 1 C1 | 2 RA | 3 RB | 4, 5 - | 6 JA | 7 JB.  Larger blocks: 0 RA | 1 RB | 2 JA | 3 JB | 4.. C.
template <int NC> struct Waves { static constexpr int N = NC == 2 ? 8 : NC + 4; };
template <int NC>
__device__ __forceinline__ int wave_role(int wave) {       // 0 RA, 1 RB, 2 JA, 3 JB, 4 + w: covariance wavefront w, -1: none
    if constexpr (NC == 2) return wave < 2 ? 4 + wave : wave < 4 ? wave - 2 : wave < 6 ? -1 : wave - 4;
    else return wave;
}
// wavefronts per SIMD the kernel is made for: those of ONE block (a CU never holds two: the launcher gives every CU at most one
// block, and from NC = 8 on a second one would not fit the LDS) -- 256 registers per lane up to NC = 4, 168 at NC = 8, 128 above
#ifndef ESKF_EP_MIN_WAVES
#define ESKF_EP_MIN_WAVES(NC) ((Waves<NC>::N + 3) / 4)
#endif
template <typename T, int NC, int PP>
__global__ __launch_bounds__(Waves<NC>::N * 64, ESKF_EP_MIN_WAVES(NC))
void eskf_epochs_kernel(EpochsArgs A, DevParams<T> prm) {
    using L = ep::Lay<T, NC, PP>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    T* lds = reinterpret_cast<T*>(smem_raw);
    int* flags = reinterpret_cast<int*>(lds + L::O_FLAGS);
    const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const int lane = (int)threadIdx.x & 63;
    if (threadIdx.x < ep::NFLAGS) flags[threadIdx.x] = 0;
    __syncthreads();
    globalise(A.a);
    A.n_pred = globalise(A.n_pred);
    A.xlog = globalise(A.xlog);
    A.stamps = globalise(A.stamps);
    const int64_t b0 = (int64_t)blockIdx.x * L::NI;
    const int ni = (int)(A.a.B - b0 < L::NI ? A.a.B - b0 : L::NI);      // the block's instances (a ragged last block has fewer)
    const int nca = (ni + 2 * PP - 1) / (2 * PP);                        // covariance wavefronts at work
    unsigned long long wait_cycles[4] = {0, 0, 0, 0};
#ifdef ESKF_X_EPSTAMPS
    const unsigned long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    // (the recurrences and the Jacobian blocks are what the covariance wavefronts wait for: they go first on their SIMDs)
    const int role = wave_role<NC>(wave);
    if (role >= 0 && role < 4) __builtin_amdgcn_s_setprio(2);
    if (role == 0) ep::role_ra<T, L>(A, prm, lds, lane, b0, ni, nca, wait_cycles);
    else if (role == 1) ep::role_rb<T, L>(A, prm, lds, lane, b0, ni, nca, wait_cycles);
    else if (role == 2) ep::role_j<T, L, 0>(A, prm, lds, lane, nca, wait_cycles);
    else if (role == 3) ep::role_j<T, L, 1>(A, prm, lds, lane, nca, wait_cycles);
    else if (role >= 4 && role - 4 < nca) ep::role_c<T, L>(A, prm, lds, role - 4, lane, b0, ni, wait_cycles);
#ifdef ESKF_X_EPSTAMPS
    if (A.stamps && (threadIdx.x & 63) == 0) {
        unsigned long long* s = A.stamps + ((size_t)blockIdx.x * 16 + (role < 0 ? 15 : role)) * 8;
        s[0] = __builtin_amdgcn_s_memtime() - t_begin;
        for (int k = 0; k < 4; ++k) s[1 + k] = wait_cycles[k];
        s[5] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// ---- host side -----------------------------------------------------------------------------------------------------
// DVI_ESKF_EPOCHS=0: runs with epochs of several predicts are enqueued epoch by epoch (the A/B, and the bitwise test of
// this kernel against that path); DVI_ESKF_EPOCHS_NC=2|4|8|10|12 pins the covariance wavefronts per block
static const int g_epochs = []() { const char* e = getenv("DVI_ESKF_EPOCHS"); return e ? atoi(e) : -1; }();
static const int g_epochs_nc = []() { const char* e = getenv("DVI_ESKF_EPOCHS_NC"); return e ? atoi(e) : 0; }();
static const int g_epochs_pp = []() { const char* e = getenv("DVI_ESKF_EPOCHS_PP"); return e ? atoi(e) : 0; }();     // 1 | 2 (fp32) pins the pairs per wavefront

static int n_compute_units() {
    static const int n_cu = []() {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        return n;
    }();
    return n_cu;
}
// Covariance wavefronts per block.  A small batch is spread over the whole chip -- the smallest block that still gives every
// CU at most one block: with two covariance wavefronts per block each has a SIMD to itself -- up to the twelve that fit next
// to the four producers.  Measured on config #5's schedule (tools/sweep_epochs_nc.sh, us per camera frame, fp32 / fp64, one
// generation of blocks): NC = 2 38 / 41, 4 49 / 52, 8 64 / 74, 10 73 / 87, 12 78 / 93 -- against 69-94 / 79-119 for the same
// batches enqueued epoch by epoch.  Beyond one generation of twelve-wavefront blocks (B > 24 CUs) the epoch-by-epoch path is
// the faster one (its lane-per-instance kernels work with all 64 lanes of a wavefront, the producers here with 2 NC of them)
// and the launcher leaves the run to it unless DVI_ESKF_EPOCHS=1 asks for this kernel; 0 = the kernel serves nothing.
// fp32, two pairs per covariance wavefront (PP = 2; 32 / 40 / 48 instances per block): the producers' lanes are used twice
// as well, the covariance wavefronts' own work is the same instructions.  One generation of 32-instance blocks takes
// 97-101 us per frame whatever its fill (B = 8 192: 2.7-2.8e9 IMU steps/s against 2.57-2.6e9 epoch by epoch), 40 / 48 per block
// 132 / 133 us (B = 10 240 / 12 288: level with the epoch-by-epoch path or behind it): the launcher takes PP = 2 where a
// generation of 32-instance blocks is at least nine tenths full, B in [29 CUs, 32 CUs].  (DVI_ESKF_EPOCHS_PP=1 | 2 pins it.)
// Returns the instances per block, 2 NC PP (PP = 2 with NC = 8 | 10 | 12, fp32 only: 32 | 40 | 48 instances).
static int epochs_block(int64_t B, bool f32) {
    const bool nc_pinned = g_epochs_nc == 2 || g_epochs_nc == 4 || g_epochs_nc == 8 || g_epochs_nc == 10 || g_epochs_nc == 12;
    const bool pp2_ok = f32 && g_epochs_pp != 1;
    if (nc_pinned) return 2 * g_epochs_nc * ((pp2_ok && g_epochs_pp == 2 && g_epochs_nc >= 8) ? 2 : 1);
    const int64_t cus = n_compute_units();
    if (g_epochs_pp != 2)
        for (int nc : {2, 4, 8, 10, 12}) if ((B + 2 * nc - 1) / (2 * nc) <= cus) return 2 * nc;
    if (pp2_ok && g_epochs_pp == 2)
        for (int nc : {8, 10, 12}) if ((B + 4 * nc - 1) / (4 * nc) <= cus) return 4 * nc;
    if (pp2_ok && B >= 29 * cus && B <= 32 * cus) return 32;
    return g_epochs == 1 ? (pp2_ok && g_epochs_pp == 2 ? 48 : 24) : 0;
}
template <typename T>
static int epochs_kernel_block_typed(int lpi, const StepArgs& a, const DevParams<T>& prm) {
    static const bool lds_ok = []() {
        int dev = 0, lim = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        if (hipDeviceGetAttribute(&lim, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) != hipSuccess) return false;
        return (size_t)lim >= ep::Lay<double, 12>::BYTES;
    }();
    const bool ok = lds_ok && g_epochs != 0 && lpi == 1 && !a.noise && !a.K_out && !a.sub_count && !prm.jfull && !prm.q_gyro &&
                    (prm.quirks & QK_Q1) && a.B > 0 && a.B < 0x7fffffffLL;
    return ok ? epochs_block(a.B, sizeof(T) == 4) : 0;
}
int epochs_kernel_block(int dtype, int lpi, const StepArgs& a, const HostParams& hp) {
    if (dtype == 0) return epochs_kernel_block_typed<double>(lpi, a, make_dev_params<double>(hp));
    return epochs_kernel_block_typed<float>(lpi, a, make_dev_params<float>(hp));
}
template <typename T, int NC, int PP = 1>
static hipError_t launch_epochs_nc(const EpochsArgs& ea, const DevParams<T>& prm, hipStream_t stream) {
    using L = ep::Lay<T, NC, PP>;
    static_assert(L::BYTES <= 160 * 1024, "the block's LDS");
    // (more than 64 KB of dynamic LDS; runtimes that do not need to be told ignore it)
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&eskf_epochs_kernel<T, NC, PP>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)L::BYTES);
    (void)attr;
    const unsigned blocks = (unsigned)((ea.a.B + L::NI - 1) / L::NI);
    hipLaunchKernelGGL((eskf_epochs_kernel<T, NC, PP>), dim3(blocks), dim3(Waves<NC>::N * 64), L::BYTES, stream, ea, prm);
    return hipGetLastError();
}
template <typename T>
static hipError_t launch_epochs_typed(int lpi, const EpochsArgs& ea, const DevParams<T>& prm, hipStream_t stream) {
    const int ni = epochs_kernel_block_typed<T>(lpi, ea.a, prm);
    if (!ni || ea.n_epochs < 1 || ea.n_total < 0) return hipErrorInvalidValue;
    static const bool verbose = []() { const char* e = getenv("DVI_ESKF_VERBOSE"); return e && e[0] == '1'; }();
    static bool said = false;
    if (verbose && !said) {
        said = true;
        fprintf(stderr, "dvi_eskf: runs of epochs go through eskf_epochs_kernel (%d instances per block)\n", ni);
    }
    switch (ni) {
        case 4: return launch_epochs_nc<T, 2>(ea, prm, stream);
        case 8: return launch_epochs_nc<T, 4>(ea, prm, stream);
        case 16: return launch_epochs_nc<T, 8>(ea, prm, stream);
        case 20: return launch_epochs_nc<T, 10>(ea, prm, stream);
        case 24: return launch_epochs_nc<T, 12>(ea, prm, stream);
        default: break;
    }
    if constexpr (sizeof(T) == 4) {
        switch (ni) {
            case 32: return launch_epochs_nc<T, 8, 2>(ea, prm, stream);
            case 40: return launch_epochs_nc<T, 10, 2>(ea, prm, stream);
            case 48: return launch_epochs_nc<T, 12, 2>(ea, prm, stream);
            default: break;
        }
    }
    return hipErrorInvalidValue;
}
// The 30-column FilterTraj rows (FilterTraj.py:12-41) of a run from its state snapshots: row r of instance b from
// xlog[r][26][n_log], the run's last row from the (flushed) state record xs[36][B]
template <typename T>
__global__ __launch_bounds__(256) void eskf_log_rows_run_kernel(int64_t n_rows, int64_t n_log, const T* xlog, const T* xs, int64_t B,
                                                                const double* t_rows, T* rows) {
    const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tid >= n_rows * n_log) return;
    const int64_t r = tid / n_log, b = tid - r * n_log;
    const bool last = r + 1 == n_rows;
    const T* src = last ? xs + b : xlog + r * 26 * n_log + b;
    const int64_t st = last ? B : n_log;
    auto X = [&](int w) { return src[(int64_t)w * st]; };
    T q[4] = {X(6), X(7), X(8), X(9)}, qc[4] = {X(22), X(23), X(24), X(25)};
    T* o = rows + tid * 30;
    const T k = T(57.29577951308232087679815481410517);
    T e[3];
    o[0] = (T)t_rows[r];
#pragma unroll
    for (int i = 0; i < 3; ++i) { o[1 + i] = X(i); o[4 + i] = X(3 + i); o[20 + i] = X(19 + i); }
    euler_xyz_deg(q, e);
#pragma unroll
    for (int i = 0; i < 3; ++i) o[7 + i] = e[i];
    o[10] = q[3]; o[11] = q[0]; o[12] = q[1]; o[13] = q[2];
#pragma unroll
    for (int i = 0; i < 3; ++i) { o[14 + i] = k * X(10 + i); o[17 + i] = X(13 + i); }
    euler_xyz_deg(qc, e);
#pragma unroll
    for (int i = 0; i < 3; ++i) o[23 + i] = e[i];
    o[26] = qc[3]; o[27] = qc[0]; o[28] = qc[1]; o[29] = qc[2];
}
hipError_t launch_log_rows_run(int dtype, int64_t n_rows, int64_t n_log, const void* xlog, const void* xs, int64_t B, const double* t_rows,
                               void* rows, hipStream_t stream) {
    if (n_rows <= 0 || n_log <= 0) return hipSuccess;
    const int64_t n = n_rows * n_log;
    if ((n + 255) / 256 > 0x7fffffffLL) return hipErrorInvalidValue;
    const unsigned blocks = (unsigned)((n + 255) / 256);
    if (dtype == 0)
        hipLaunchKernelGGL(eskf_log_rows_run_kernel<double>, dim3(blocks), dim3(256), 0, stream, n_rows, n_log, (const double*)xlog, (const double*)xs, B,
                           t_rows, (double*)rows);
    else
        hipLaunchKernelGGL(eskf_log_rows_run_kernel<float>, dim3(blocks), dim3(256), 0, stream, n_rows, n_log, (const float*)xlog, (const float*)xs, B,
                           t_rows, (float*)rows);
    return hipGetLastError();
}

hipError_t launch_epochs(int dtype, int lpi, const EpochsArgs& ea, const HostParams& hp, hipStream_t stream) {
    if (dtype == 0) return launch_epochs_typed<double>(lpi, ea, make_dev_params<double>(hp), stream);
    return launch_epochs_typed<float>(lpi, ea, make_dev_params<float>(hp), stream);
}

}  // namespace eskf

// opensot_amd/csrc/osot_kin.h -- batched kinematics producer.
//
// One wavefront per instance, LANE = JOINT (n <= 64).  Per instance:
//   1. lane j: local transform T_j(q_j) = [R0_j Rot(axis_j, q_j) | p0_j] (revolute) or [R0_j | p0_j + R0_j axis_j q_j]
//      (prismatic) -> LDS
//   2. lane j: world transform of its joint frame by pointer jumping over the ancestor chain (log2(depth) rounds)
//      -> LDS, with the world axis z_j and origin p_j
//   3. frames: pose out; column j of the frame Jacobian = [z_j x (p_f - p_j); z_j] (revolute) / [z_j; 0] (prismatic)
//      if joint j is an ancestor of the frame's link, else 0.  Row r of the 6 x n block is written by all lanes at
//      once: one coalesced 8 n-byte store per row, straight into the stacked A_k.
//   4. centre of mass: c = sum m_l c_l / M (wave reduction); column j of its Jacobian from the subtree aggregates
//      (mass and first moment of the links joint j moves): z_j x (Sc_j - Sm_j p_j) / M  (prismatic: (Sm_j / M) z_j)
//   5. self-collision pairs (capsule / sphere pairs): lane p = pair for the closest points of the two axis segments
//      (world frame) -> distance d = |c_a - c_b| - r_a - r_b and unit normal n -> LDS; then one coalesced row per
//      pair, lane j = joint: J_d[j] = n . (v_j(c_a) [j moves a] - v_j(c_b) [j moves b]),  v_j(c) = z_j x (c - p_j)
//      (revolute) or z_j (prismatic).  (A point of the capsule SURFACE, c - r n, has the same velocity along n.)
// HBM-bound by construction: reads 8 n bytes of q, writes (6 F + 3) n 8 + 96 F + 24 bytes per instance.
#pragma once
#include <osot_team.h>
#include <osot_mi355x.h>

namespace osot {

struct DevKin {            // osot_kin_desc + ancestor masks, in device memory
    osot_kin_desc d;
    unsigned long long anc[OSOT_KIN_MAX_JOINTS];   // bit a set: joint a is an ancestor of (or is) joint j
    unsigned long long sub[OSOT_KIN_MAX_JOINTS];   // bit l set: link l is moved by joint j (j itself included)
    double total_mass;
    // depth-first (pre-order) position of every joint: the links a joint moves are the positions dfs_pos[j] .. sub_end[j] - 1, so a
    // subtree aggregate is a DIFFERENCE OF PREFIX SUMS over that order (the centre-of-mass Jacobian, stage 4)
    int dfs_pos[OSOT_KIN_MAX_JOINTS];
    int sub_end[OSOT_KIN_MAX_JOINTS];
};

// the tables derived from the tree (h.d set, parent[j] < j checked by the caller): shared by osot_kin_create and the emulator's driver
inline void kin_build_tables(DevKin& h) {
    const int n = h.d.n;
    h.total_mass = 0.0;
    for (int j = 0; j < OSOT_KIN_MAX_JOINTS; ++j) { h.anc[j] = 0ull; h.sub[j] = 0ull; h.dfs_pos[j] = j; h.sub_end[j] = j + 1; }
    for (int j = 0; j < n; ++j) {
        h.anc[j] = (1ull << j) | (h.d.parent[j] >= 0 ? h.anc[h.d.parent[j]] : 0ull);
        for (int a = 0; a <= j; ++a) if ((h.anc[j] >> a) & 1ull) h.sub[a] |= (1ull << j);
        h.total_mass += h.d.mass[j];
    }
    if (!(h.total_mass > 0.0)) h.total_mass = 1.0;
    int next_child[OSOT_KIN_MAX_JOINTS], root_off = 0;      // parents come before their children: one pass
    for (int j = 0; j < n; ++j) {
        const int size = __builtin_popcountll(h.sub[j]), pa = h.d.parent[j];
        if (pa < 0) { h.dfs_pos[j] = root_off; root_off += size; }
        else { h.dfs_pos[j] = next_child[pa]; next_child[pa] += size; }
        next_child[j] = h.dfs_pos[j] + 1;
        h.sub_end[j] = h.dfs_pos[j] + size;
    }
}

__device__ __forceinline__ void mat3_mul(const double* A, const double* B, double* C) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) C[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
}
__device__ __forceinline__ void mat3_vec(const double* A, const double* v, double* o) {
#pragma unroll
    for (int i = 0; i < 3; ++i) o[i] = A[3 * i] * v[0] + A[3 * i + 1] * v[1] + A[3 * i + 2] * v[2];
}
__device__ __forceinline__ void cross3(const double* a, const double* b, double* o) {
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

__device__ __forceinline__ double clamp01(double v) { return v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v); }
// closest points of the segments [p1, q1] and [p2, q2] (either may have zero length): parameters s, t in [0, 1]
__device__ inline void closest_segment_points(const double* p1, const double* q1, const double* p2, const double* q2,
                                              double* c1, double* c2) {
    const double tiny = 1.0e-18;
    double d1[3], d2[3], r[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { d1[i] = q1[i] - p1[i]; d2[i] = q2[i] - p2[i]; r[i] = p1[i] - p2[i]; }
    const double a = d1[0] * d1[0] + d1[1] * d1[1] + d1[2] * d1[2];
    const double e = d2[0] * d2[0] + d2[1] * d2[1] + d2[2] * d2[2];
    const double f = d2[0] * r[0] + d2[1] * r[1] + d2[2] * r[2];
    const double c = d1[0] * r[0] + d1[1] * r[1] + d1[2] * r[2];
    double s = 0.0, t = 0.0;
    if (a <= tiny && e <= tiny) {
    } else if (a <= tiny) {
        t = clamp01(f / e);
    } else if (e <= tiny) {
        s = clamp01(-c / a);
    } else {
        const double b = d1[0] * d2[0] + d1[1] * d2[1] + d1[2] * d2[2];
        const double den = a * e - b * b;            // >= 0; 0 for parallel axes: any s, take 0
        s = (den > tiny * a * e) ? clamp01((b * f - c * e) / den) : 0.0;
        t = (b * s + f) / e;
        if (t < 0.0) { t = 0.0; s = clamp01(-c / a); }
        else if (t > 1.0) { t = 1.0; s = clamp01((b - c) / a); }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) { c1[i] = p1[i] + s * d1[i]; c2[i] = p2[i] + t * d2[i]; }
}

// closest points of the segment [p0, p1] and the axis-aligned box [-h, h] (all in the BOX frame): ca on the segment, cb on
// the box.  dist^2(t) = sum_i max(|P_i(t)| - h_i, 0)^2 along P(t) = p0 + t (p1 - p0) is convex and piecewise quadratic with
// at most six breakpoints (where a coordinate crosses one of its two faces): on every piece the minimiser of the quadratic
// is clamped to the piece and the best piece wins -- exact, no iteration (FCL, which the reference's collision module
// wraps, runs GJK on the same convex problem).
__device__ inline void closest_segment_box(const double* p0, const double* p1, const double* h, double* ca, double* cb) {
    double v[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    double bp[8];
    bp[0] = 0.0; bp[7] = 1.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const bool mv = fabs(v[i]) > 1.0e-300;
        const double ta = mv ? (-h[i] - p0[i]) / v[i] : 0.0, tb = mv ? (h[i] - p0[i]) / v[i] : 0.0;
        bp[1 + 2 * i] = clamp01(ta);
        bp[2 + 2 * i] = clamp01(tb);
    }
    // sort the eight values (a fixed odd-even network: no data-dependent indexing, the array stays in registers)
#pragma unroll
    for (int pass = 0; pass < 8; ++pass) {
#pragma unroll
        for (int i = (pass & 1); i + 1 < 8; i += 2) {
            const double lo = fmin(bp[i], bp[i + 1]), hi = fmax(bp[i], bp[i + 1]);
            bp[i] = lo; bp[i + 1] = hi;
        }
    }
    double best = INFINITY, tbest = 0.0;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
        const double t0 = bp[k], t1 = bp[k + 1], tm = 0.5 * (t0 + t1);
        double qa = 0.0, qb = 0.0;          // f(t) = qa t^2 + 2 qb t + qc on this piece; minimiser -qb / qa
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double pm = p0[i] + tm * v[i];
            const double side = pm > h[i] ? 1.0 : (pm < -h[i] ? -1.0 : 0.0);
            const double off = p0[i] - side * h[i];      // excess_i(t) = off + t v_i where the coordinate is outside
            qa += (side != 0.0) ? v[i] * v[i] : 0.0;
            qb += (side != 0.0) ? off * v[i] : 0.0;
        }
        double t = (qa > 0.0) ? -qb / qa : t0;
        t = t < t0 ? t0 : (t > t1 ? t1 : t);
        double f = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double pt = p0[i] + t * v[i];
            const double ex = pt > h[i] ? pt - h[i] : (pt < -h[i] ? pt + h[i] : 0.0);
            f += ex * ex;
        }
        if (f < best) { best = f; tbest = t; }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        ca[i] = p0[i] + tbest * v[i];
        cb[i] = ca[i] > h[i] ? h[i] : (ca[i] < -h[i] ? -h[i] : ca[i]);
    }
}

#ifndef OSOT_KIN_TS
#define OSOT_KIN_TS 12
#endif
// LDS one instance of kin_instance needs, in doubles (the int / 64-bit tables included): transforms, axes, centres of mass, parents,
// ancestor masks -- and the pair table of the PAIRS instantiation
template <int JMAX> constexpr int kin_lds_doubles(bool pairs) {
    return JMAX * OSOT_KIN_TS + JMAX * 3 + JMAX * 4 + (JMAX + 1) / 2 + JMAX + OSOT_KIN_MAX_FRAMES * (14 + 1) + (pairs ? OSOT_KIN_MAX_PAIRS * 9 : 0);
}
// The kinematics of ONE instance by the JMAX lanes j = 0 .. JMAX - 1 that call it together (lane = joint; the reductions are
// colsum<JMAX>).  `lds` = kin_lds_doubles<JMAX>(PAIRS) doubles of this instance's own.  The kernel below calls it for one instance
// per wavefront (JMAX = 64) or two (JMAX = 32: the halves); osot_control_cycle_kernel calls it in front of the instance's update.
// PAIRS: the instantiation with the self-collision stage (step 5); the other one keeps the leaner register / LDS budget
// developer knob (tools/build_variant.sh NAME -DOSOT_KIN_PHASES): instance 0 prints the clock count of every stage
#ifdef OSOT_KIN_PHASES
#define KIN_PHASE(tag) do { const long long t_ = (long long)clock64(); if (inst == 0 && j == 0) printf("KIN " tag " %lld\n", t_ - kph_); kph_ = (long long)clock64(); } while (0)
#else
#define KIN_PHASE(tag) do { } while (0)
#endif
template <bool PAIRS, int JMAX>
__device__ __forceinline__ void kin_instance(const DevKin* __restrict__ K, const osot_kin_batch& Bt, const long long inst, const bool live,
                                             const int j, double* lds) {
    constexpr int TS = OSOT_KIN_TS;
    double* Tb = lds;                                     // ONE transform buffer [R | p] per joint (round 3; it was a ping-pong
    //                                              pair: a round of the pointer jumping reads, synchronises, writes, synchronises
    //                                              either way, so the second buffer bought nothing and cost 3 KB per instance:
    //                                              10.4 KB per wavefront instead of 16.5 -> 15 wavefronts per CU instead of 9)
    double* Tl = Tb;
    double* Zw = Tb + JMAX * TS;                          // world joint axes
    double* Cw = Zw + JMAX * 3;                           // world link centres of mass, mass
    unsigned long long* Anc = reinterpret_cast<unsigned long long*>(Cw + JMAX * 4);   // ancestor masks
    int* Par = reinterpret_cast<int*>(Anc + JMAX);        // parent indices (the chain walk must not chase pointers through HBM)
    double* Fw = reinterpret_cast<double*>(Par + 2 * ((JMAX + 1) / 2));               // per frame: world [R | p], its joint, body flag
    unsigned long long* Fq = reinterpret_cast<unsigned long long*>(Fw + OSOT_KIN_MAX_FRAMES * 14);   // ... column mask
    double* Pw = reinterpret_cast<double*>(Fq + OSOT_KIN_MAX_FRAMES);                 // per pair: normal n, axis points c_a, c_b (world)
    (void)Pw;
#ifdef OSOT_KIN_PHASES
    long long kph_ = (long long)clock64();
#endif
    const int n = K->d.n;
    const bool valid = j < n && live;
    Par[j] = valid ? K->d.parent[j] : -1;
    Anc[j] = valid ? K->anc[j] : 0ull;
    // ---- 1. local transforms.  Everything the lane needs from the model (23 loads) and its q are fetched in ONE batch, in front of the
    // sincos (computed whatever the joint type: its ~1.5 k clocks then run under the loads instead of behind the type's round trip)
    const int jc = (j < n) ? j : 0;
    const double qj = Bt.q[(live ? inst : 0) * n + jc];
    const int type_j = K->d.type[jc];
    const int dfs_j = K->dfs_pos[jc], end_j = K->sub_end[jc];
    double R0[9], ax[3], p0j[3], comj[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) R0[i] = K->d.R0[jc][i];
#pragma unroll
    for (int i = 0; i < 3; ++i) { ax[i] = K->d.axis[jc][i]; p0j[i] = K->d.p0[jc][i]; comj[i] = K->d.com[jc][i]; }
    const double mass_j = K->d.mass[jc];
    {
        double sn, cs;
        sincos(qj, &sn, &cs);
        double R[9], p[3];
        if (type_j == OSOT_JOINT_REVOLUTE) {
            const double v = 1.0 - cs, x = ax[0], y = ax[1], z = ax[2];
            const double Rq[9] = {cs + x * x * v,     x * y * v - z * sn, x * z * v + y * sn,
                                  y * x * v + z * sn, cs + y * y * v,     y * z * v - x * sn,
                                  z * x * v - y * sn, z * y * v + x * sn, cs + z * z * v};   // Rodrigues
            mat3_mul(R0, Rq, R);
#pragma unroll
            for (int i = 0; i < 3; ++i) p[i] = p0j[i];
        } else {
            double t[3];
            mat3_vec(R0, ax, t);
#pragma unroll
            for (int i = 0; i < 9; ++i) R[i] = R0[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) p[i] = p0j[i] + qj * t[i];
        }
        if (valid) {
#pragma unroll
            for (int i = 0; i < 9; ++i) Tl[j * TS + i] = R[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) Tl[j * TS + 9 + i] = p[i];
        }
    }
    wave_sync();
    KIN_PHASE("local");
    // ---- 2. world transforms by POINTER JUMPING: every lane keeps T(jp -> j), the transform from the frame of its
    // jump pointer jp to its own frame, and in each round composes it with T(jp(jp) -> jp) read from LDS and jumps:
    // ceil(log2(depth + 1)) rounds (4 for a humanoid) instead of a walk of `depth` steps up the chain.
    double Rw[9], pw[3];
#pragma unroll
    for (int i = 0; i < 9; ++i) Rw[i] = valid ? Tl[j * TS + i] : 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) pw[i] = valid ? Tl[j * TS + 9 + i] : 0.0;
    {
        int jp = Par[j];
        while (wave_ballot(jp >= 0) != 0ull) {
            int njp = jp;
            if (jp >= 0) {
                double Ra[9], pa[3], Rn[9], pn[3];
#pragma unroll
                for (int i = 0; i < 9; ++i) Ra[i] = Tb[jp * TS + i];
#pragma unroll
                for (int i = 0; i < 3; ++i) pa[i] = Tb[jp * TS + 9 + i];
                mat3_mul(Ra, Rw, Rn);
                mat3_vec(Ra, pw, pn);
#pragma unroll
                for (int i = 0; i < 9; ++i) Rw[i] = Rn[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) pw[i] = pn[i] + pa[i];
                njp = Par[jp];
            }
            wave_sync();            // everybody has read Par and the transforms of this round
#pragma unroll
            for (int i = 0; i < 9; ++i) Tb[j * TS + i] = Rw[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) Tb[j * TS + 9 + i] = pw[i];
            Par[j] = njp;
            jp = njp;
            wave_sync();
        }
    }
    KIN_PHASE("jump");
    double* Tw = Tb;               // final world transforms (re-written below, after the last round's barrier: a joint without a parent never entered the loop's writes)
    double zj[3];
    {
        double z[3], cl[3];
        mat3_vec(Rw, ax, z);
        mat3_vec(Rw, comj, cl);
        if (valid) {
#pragma unroll
            for (int i = 0; i < 9; ++i) Tw[j * TS + i] = Rw[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) Tw[j * TS + 9 + i] = pw[i];
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) { Zw[j * 3 + i] = valid ? z[i] : 0.0; zj[i] = valid ? z[i] : 0.0; }
        // m_j [c_j, 1] at the joint's depth-first position (joints beyond n: zeros at their own index, which no joint below n has)
        const int pos = valid ? dfs_j : j;
#pragma unroll
        for (int i = 0; i < 3; ++i) Cw[pos * 4 + i] = valid ? mass_j * (cl[i] + pw[i]) : 0.0;
        Cw[pos * 4 + 3] = valid ? mass_j : 0.0;
    }
    const bool revolute = valid && type_j == OSOT_JOINT_REVOLUTE;
    // ---- 3a. world frames, lane = frame: the frame's joint, offset and options in one batch of loads per lane, [R | p] and the joint
    // into LDS for the Jacobian columns below; the pose goes out from here
    const int nfr = K->d.n_frames;
    {
        const int f = (j < nfr) ? j : 0;
        const int jf = K->d.frame_joint[f];
        double FR[9], Fp[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) FR[i] = K->d.frame_R[f][i];
#pragma unroll
        for (int i = 0; i < 3; ++i) Fp[i] = K->d.frame_p[f][i];
        const int body_f = K->d.frame_body[f];
        const int base_f = K->d.frame_base[f];                    // 0: the world; g + 1: frame g is the base link frame (Cartesian.cpp:73-81)
        const unsigned long long cm_f = K->d.frame_col_mask[f];   // Task::applyActiveJointsMask (Task.h:129-139)
        double* pose_out = nullptr;        // (a select chain: a lane-indexed read of the argument struct would be served from a scratch copy)
#pragma unroll
        for (int ff = 0; ff < OSOT_KIN_MAX_FRAMES; ++ff) pose_out = (f == ff) ? Bt.frame_pose[ff] : pose_out;
        wave_sync();               // the world transforms are in LDS
        double Rj[9], pj[3], Rf[9], t[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) Rj[i] = Tw[jf * TS + i];
#pragma unroll
        for (int i = 0; i < 3; ++i) pj[i] = Tw[jf * TS + 9 + i];
        mat3_mul(Rj, FR, Rf);
        mat3_vec(Rj, Fp, t);
        if (j < nfr) {
#pragma unroll
            for (int i = 0; i < 9; ++i) Fw[j * 14 + i] = Rf[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) Fw[j * 14 + 9 + i] = pj[i] + t[i];
            Fw[j * 14 + 12] = (double)jf;
            Fw[j * 14 + 13] = (double)((body_f ? 1 : 0) + 2 * base_f);   // options: bit 0 = BODY Jacobian, the rest = base frame + 1
            Fq[j] = cm_f;
            if (pose_out && live && base_f <= 0) {
#pragma unroll
                for (int i = 0; i < 9; ++i) pose_out[inst * 12 + i] = Rf[i];
#pragma unroll
                for (int i = 0; i < 3; ++i) pose_out[inst * 12 + 9 + i] = pj[i] + t[i];
            }
        }
        wave_sync();
        if (j < nfr && pose_out && live && base_f > 0) {
            // relative pose base_T_distal = [R_b'R_d | R_b'(p_d - p_b)] (ModelInterface::getPose(distal, base, T), Cartesian.cpp:80-81);
            // the base frame's world pose was published by its own lane above
            const double* Bw = Fw + (base_f - 1) * 14;
            double Rb[9], dp[3];
#pragma unroll
            for (int i = 0; i < 9; ++i) Rb[i] = Bw[i];
#pragma unroll
            for (int i = 0; i < 3; ++i) dp[i] = (pj[i] + t[i]) - Bw[9 + i];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int cc = 0; cc < 3; ++cc)
                    pose_out[inst * 12 + 3 * r + cc] = Rb[r] * Rf[cc] + Rb[3 + r] * Rf[3 + cc] + Rb[6 + r] * Rf[6 + cc];
                pose_out[inst * 12 + 9 + r] = Rb[r] * dp[0] + Rb[3 + r] * dp[1] + Rb[6 + r] * dp[2];
            }
        }
    }
    KIN_PHASE("world");
    // ---- 3. frames: the Jacobian columns, lane = joint
    // (a frame's transform and options come from the LDS table in one batch of reads -- fetched from the model here, every frame started
    //  behind a round trip of scalar loads; the row address stays a scalar load of the argument struct: through LDS it became a
    //  per-lane address and the six stores paid for it, 2.9 k -> 3.4 k clocks)
    for (int f = 0; f < nfr; ++f) {
        double Rf[9], pf[3];
#pragma unroll
        for (int i = 0; i < 9; ++i) Rf[i] = Fw[f * 14 + i];
#pragma unroll
        for (int i = 0; i < 3; ++i) pf[i] = Fw[f * 14 + 9 + i];
        const int jf = (int)Fw[f * 14 + 12];
        const int opt = (int)Fw[f * 14 + 13];
        const bool body = (opt & 1) != 0;
        const int bf = (opt >> 1) - 1;                   // base frame of a relative Cartesian task, -1: the world
        const int jb = (bf >= 0) ? (int)Fw[bf * 14 + 12] : 0;
        const unsigned long long cm = Fq[f];
        double* Jf = Bt.frame_J[f];                      // (the argument struct: scalar loads, a uniform base address for the stores)
        if (!Jf) continue;
        const long long jstride = Bt.frame_J_stride[f];
        if (valid) {
            double col[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
            // joint j moves the distal link (+1), the base link (-1: the base body's point at the distal origin moves the same way,
            // so the joints both chains share cancel), both or neither (0).  World base: the ancestor test as before
            const bool in_d = ((Anc[jf] >> j) & 1ull) != 0ull;
            const bool in_b = bf >= 0 && ((Anc[jb] >> j) & 1ull) != 0ull;
            if (in_d != in_b) {
                if (revolute) {
                    const double dlt[3] = {pf[0] - pw[0], pf[1] - pw[1], pf[2] - pw[2]};
                    cross3(zj, dlt, col);
                    col[3] = zj[0]; col[4] = zj[1]; col[5] = zj[2];
                } else {
                    col[0] = zj[0]; col[1] = zj[1]; col[2] = zj[2];
                }
                if (in_b) {
#pragma unroll
                    for (int i = 0; i < 6; ++i) col[i] = -col[i];
                }
            }
            if (body || bf >= 0) {
                // BODY Jacobian Ad(R_f') J (Cartesian.cpp:93-100): both halves rotated by R_f'; a relative Jacobian is expressed in
                // its base frame: R_b' (getRelativeJacobian, Cartesian.cpp:75-76) -- and BODY on top of that is Ad((R_b'R_d)') R_b' = R_d'
                double Rr[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) Rr[i] = body ? Rf[i] : Fw[(bf >= 0 ? bf : 0) * 14 + i];
                double rl[3], ra[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    rl[i] = Rr[i] * col[0] + Rr[3 + i] * col[1] + Rr[6 + i] * col[2];
                    ra[i] = Rr[i] * col[3] + Rr[3 + i] * col[4] + Rr[6 + i] * col[5];
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) { col[i] = rl[i]; col[3 + i] = ra[i]; }
            }
            const bool masked = cm != 0ull && !((cm >> j) & 1ull);
            double* J = Jf + inst * jstride;
#pragma unroll
            for (int r = 0; r < 6; ++r) J[r * n + j] = masked ? 0.0 : col[r];
        }
    }
    KIN_PHASE("frames");
    // ---- 4. centre of mass and its Jacobian
    if (Bt.com || Bt.com_J) {
        const double iM = fast_rcp(K->total_mass);       // (one reciprocal instead of six fp64 divisions)
        // Inclusive prefix sums of m_l [c_l, 1] over the depth-first order (written there by the world stage), lane = position:
        // log2(JMAX) rounds of "read the entry d positions back, add, write".  The links joint j moves are a contiguous range of that
        // order, so S_j = P[sub_end_j - 1] - P[dfs_j - 1]; the total is the last entry (positions beyond n hold zeros).
        // (the 0/1 matrix-vector product this replaces read all 4 JMAX entries per lane: 4.4 k of the producer's 15.9 k clocks)
        double a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = Cw[j * 4 + i];
#pragma unroll
        for (int d = 1; d < JMAX; d <<= 1) {
            double t[4];
            const int src = (j >= d) ? j - d : 0;
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = Cw[src * 4 + i];
            wave_sync();
#pragma unroll
            for (int i = 0; i < 4; ++i) { a[i] += (j >= d) ? t[i] : 0.0; Cw[j * 4 + i] = a[i]; }
            wave_sync();
        }
        if (Bt.com) {
            const int i3 = (j < 3) ? j : 0;
            const double tot = Cw[(JMAX - 1) * 4 + i3];
            if (j < 3 && live) Bt.com[inst * 3 + j] = tot * iM;
        }
        if (Bt.com_J) {
            // column j = z_j x (Sc_j - Sm_j p_j) / M  (revolute)   or   (Sm_j / M) z_j  (prismatic)
            const int hi = valid ? end_j - 1 : 0, lo = (valid && dfs_j > 0) ? dfs_j - 1 : 0;
            double sc[3], sm;
            {
                double ph[4], pl[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) { ph[i] = Cw[hi * 4 + i]; pl[i] = Cw[lo * 4 + i]; }
                const bool from0 = !(valid && dfs_j > 0);
#pragma unroll
                for (int i = 0; i < 3; ++i) sc[i] = ph[i] - (from0 ? 0.0 : pl[i]);
                sm = ph[3] - (from0 ? 0.0 : pl[3]);
            }
            double acc[3];
            if (revolute) {
                const double dlt[3] = {sc[0] - sm * pw[0], sc[1] - sm * pw[1], sc[2] - sm * pw[2]};
                cross3(zj, dlt, acc);
            } else {
                acc[0] = sm * zj[0]; acc[1] = sm * zj[1]; acc[2] = sm * zj[2];
            }
            if (valid) {
                const unsigned long long cm = K->d.com_col_mask;
                const bool masked = cm != 0ull && !((cm >> j) & 1ull);
                double* J = Bt.com_J + inst * Bt.com_J_stride;
#pragma unroll
                for (int r = 0; r < 3; ++r) J[r * n + j] = masked ? 0.0 : acc[r] * iM;
            }
        }
    }
    KIN_PHASE("com");
    // ---- 5. self-collision pairs
    if constexpr (PAIRS) {
    const int np = K->d.n_pairs;
    if (np > 0 && (Bt.pair_dist || Bt.pair_J)) {
        if (j < np && live) {
            double e[2][6];
            double Rc[9], pc[3];      // carrier frame of side b (link, world, or the runtime pose of an environment shape)
#pragma unroll
            for (int sd = 0; sd < 2; ++sd) {
                const int js = K->d.pair_joint[j][sd];
                double Rj[9], pj[3];
                const int env = (sd == 1) ? K->d.pair_env[j] - 1 : -1;
                if (js >= 0) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) Rj[i] = Tw[js * TS + i];
#pragma unroll
                    for (int i = 0; i < 3; ++i) pj[i] = Tw[js * TS + 9 + i];
                } else if (env >= 0 && Bt.env_pose) {
                    const double* E = Bt.env_pose + inst * Bt.env_pose_stride + env * 12;
#pragma unroll
                    for (int i = 0; i < 9; ++i) Rj[i] = E[i];
#pragma unroll
                    for (int i = 0; i < 3; ++i) pj[i] = E[9 + i];
                } else {
#pragma unroll
                    for (int i = 0; i < 9; ++i) Rj[i] = (i % 4 == 0) ? 1.0 : 0.0;
#pragma unroll
                    for (int i = 0; i < 3; ++i) pj[i] = 0.0;
                }
                double t0[3], t1[3];
                mat3_vec(Rj, &K->d.pair_seg[j][sd][0], t0);
                mat3_vec(Rj, &K->d.pair_seg[j][sd][3], t1);
#pragma unroll
                for (int i = 0; i < 3; ++i) { e[sd][i] = t0[i] + pj[i]; e[sd][3 + i] = t1[i] + pj[i]; }
                if (sd == 1) {
#pragma unroll
                    for (int i = 0; i < 9; ++i) Rc[i] = Rj[i];
#pragma unroll
                    for (int i = 0; i < 3; ++i) pc[i] = pj[i];
                }
            }
            double ca[3], cb[3];
            if (K->d.pair_kind[j] == OSOT_SHAPE_BOX) {
                // box frame in the world: (Rc, pc) o (shape_R, shape_p); side a's segment goes into it, the closest points
                // come back out
                double Rb[9], tb[3], pb[3];
                mat3_mul(Rc, K->d.pair_shape_R[j], Rb);
                mat3_vec(Rc, K->d.pair_shape_p[j], tb);
#pragma unroll
                for (int i = 0; i < 3; ++i) pb[i] = pc[i] + tb[i];
                double s0[3], s1[3], la[3], lb[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) {   // R_b' (x - p_b)
                    s0[i] = Rb[i] * (e[0][0] - pb[0]) + Rb[3 + i] * (e[0][1] - pb[1]) + Rb[6 + i] * (e[0][2] - pb[2]);
                    s1[i] = Rb[i] * (e[0][3] - pb[0]) + Rb[3 + i] * (e[0][4] - pb[1]) + Rb[6 + i] * (e[0][5] - pb[2]);
                }
                closest_segment_box(s0, s1, K->d.pair_box[j], la, lb);
                double wa[3], wb[3];
                mat3_vec(Rb, la, wa);
                mat3_vec(Rb, lb, wb);
#pragma unroll
                for (int i = 0; i < 3; ++i) { ca[i] = wa[i] + pb[i]; cb[i] = wb[i] + pb[i]; }
            } else {
                closest_segment_points(&e[0][0], &e[0][3], &e[1][0], &e[1][3], ca, cb);
            }
            const double dv[3] = {ca[0] - cb[0], ca[1] - cb[1], ca[2] - cb[2]};
            const double len = sqrt(dv[0] * dv[0] + dv[1] * dv[1] + dv[2] * dv[2]);
            const bool deg = !(len > 1.0e-12);      // coincident axis points: no direction; the row is zero
            const double il = deg ? 0.0 : 1.0 / len;
#pragma unroll
            for (int i = 0; i < 3; ++i) { Pw[j * 9 + i] = dv[i] * il; Pw[j * 9 + 3 + i] = ca[i]; Pw[j * 9 + 6 + i] = cb[i]; }
            if (Bt.pair_dist) Bt.pair_dist[inst * np + j] = len - K->d.pair_radius[j][0] - K->d.pair_radius[j][1];
        }
        wave_sync();
        if (Bt.pair_J) {
            double* J = Bt.pair_J + inst * Bt.pair_J_stride;
            for (int p = 0; p < np; ++p) {
                const int ja = K->d.pair_joint[p][0], jb = K->d.pair_joint[p][1];
                const double sa = ((Anc[ja] >> j) & 1ull) ? 1.0 : 0.0;
                const double sb = (jb >= 0 && ((Anc[jb < 0 ? 0 : jb] >> j) & 1ull)) ? 1.0 : 0.0;   // (a world shape: no joint moves it)
                const double nn[3] = {Pw[p * 9], Pw[p * 9 + 1], Pw[p * 9 + 2]};
                double val;
                if (revolute) {
                    // n . (z x (c_a - p_j)) sa - n . (z x (c_b - p_j)) sb = n . (z x (sa (c_a - p_j) - sb (c_b - p_j)))
                    const double dl[3] = {sa * (Pw[p * 9 + 3] - pw[0]) - sb * (Pw[p * 9 + 6] - pw[0]),
                                          sa * (Pw[p * 9 + 4] - pw[1]) - sb * (Pw[p * 9 + 7] - pw[1]),
                                          sa * (Pw[p * 9 + 5] - pw[2]) - sb * (Pw[p * 9 + 8] - pw[2])};
                    double cr[3];
                    cross3(zj, dl, cr);
                    val = nn[0] * cr[0] + nn[1] * cr[1] + nn[2] * cr[2];
                } else {
                    val = (sa - sb) * (nn[0] * zj[0] + nn[1] * zj[1] + nn[2] * zj[2]);
                }
                if (valid) J[p * n + j] = val;
            }
        }
    }
    }
}

// JMAX = 32: TWO instances per wavefront (lanes 0..31 and 32..63 each run a robot of <= 32 joints: every instruction, every
// 64-lane store carries two instances);  JMAX = 64: one instance per wavefront.
template <bool PAIRS, int JMAX>
__global__ void __launch_bounds__(64) osot_kin_kernel(const DevKin* __restrict__ K, const osot_kin_batch Bt) {
    constexpr int PACK = 64 / JMAX;
    constexpr int SLICE = kin_lds_doubles<JMAX>(PAIRS);
    OSOT_STATIC_LDS(double, kin_lds, PACK * SLICE);
    const int sub = (PACK == 2) ? (int)(threadIdx.x >> 5) : 0;
    const int j = (PACK == 2) ? (int)(threadIdx.x & 31u) : (int)threadIdx.x;
    const long long inst = (long long)blockIdx.x * PACK + sub;
    const bool live = inst < Bt.B;            // (an odd batch leaves the last wavefront's second half idle)
    kin_instance<PAIRS, JMAX>(K, Bt, inst, live, j, kin_lds + sub * SLICE);
}

}  // namespace osot

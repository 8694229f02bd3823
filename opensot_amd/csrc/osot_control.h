// opensot_amd/csrc/osot_control.h -- the reference's control-loop body for B robots in ONE launch
// (examples/cpp/coman_ik.cpp:186-219: `model.update(); stack->update(); solver->solve(dq); q = model.sum(q, dq)`):
// per instance the same wavefront runs the kinematics producer (osot_kin.h: frame poses, Jacobian rows written straight into
// A_k, CoM), AutoStack::update() and the iHQP cascade (osot_kernels.h), and integrates q += dq.
//
// Why one launch: as three launches per step (kinematics, cycle, the integration) every step of a lane carries two more
// dependent launch gaps, and -- measured, tools/full_cycle_trace.sh -- the kinematics kernel of one lane takes 57-77 us instead
// of 17 because the other lane's cascade holds every wavefront slot of the chip while it waits (BASELINE config 3's stack on the
// 32-DoF humanoid, 4096 robots in closed loop: 25.1 M solves/s as three launches, 26.2 M as one; update + cascade ALONE on the same
// problems: 30.6 M).  Here the kinematics of an instance is ~10 % more work in front of its own update: no slots to wait for, no
// gaps, and the rows it writes come back from the CU's own L1 / L2 lines.  With sub-batches on several streams it pays 4-8 % on the
// reference's COMAN stacks as well.
// All three lane layouts (32, 56, 64): the reference's own 35-coordinate COMAN runs its loop body as one launch too.
#pragma once
#include <osot_kernels.h>
#include <osot_kin.h>

namespace osot {

struct DevControl {
    const DevKin* K;           // the model (osot_kin_create)
    osot_kin_batch Bt;         // where q is read and the poses / Jacobian rows / CoM are written (the leaf inputs and A_k point there)
    double* q_int;             // [B][n] integrated at the end: q_int += dq (null: not integrated).  Usually Bt.q itself.
    // round 5 -- a ROLLOUT: `steps` control cycles of every robot in this one launch (osot_control_rollout; 0 or 1: one cycle)
    int steps;
    double* dq_steps;          // [steps][B][n] every cycle's dq, or null (D.dq holds the last cycle's either way)
    int* status_steps;         // [steps][B] every cycle's status, or null (D.status: the first non-zero status of the rollout)
    int K_pairs;               // HOST side only: the model's collision-pair count (ihqp_launch: pair outputs need a PAIRS instantiation)
};

// LDS in front of the cycle's own use of the slice (the update stages its arguments there afterwards).  NP = 32: two kinematics
// slices, the second half of the wavefront runs as an idle instance (kin_instance<.,32> is written for two robots per wavefront;
// its lanes store their -- masked -- tables unconditionally); the 64-lane kernels: one slice, lane = joint over the whole wavefront
// pairs: the instantiations whose producer carries the collision-pair stage (every one but BOX: a plan with collision rows has
// inequality rows), with the pair table in the slice
constexpr size_t control_kin_lds_bytes(int NP, bool pairs = true) {
    return NP == 32 ? 2 * sizeof(double) * (size_t)kin_lds_doubles<32>(pairs) : sizeof(double) * (size_t)kin_lds_doubles<64>(pairs);
}

// ROLL: the instantiation that carries the rollout loop (osot_control_rollout); the single-cycle kernel does not (measured: the loop
// around the body -- its trip count a run-time value -- cost the one-cycle launch 2-5 %: 26.1 -> 24.8 M on the full cycle, 20.1 -> 19.0 M
// on COMAN35 S1, A/B on one box)
template <int NP, bool EXTRA = false, bool BOX = false, bool ROLL = false>
__global__ void __launch_bounds__(64, (NP == 32 ? OSOT_WAVES32 : (NP == 40 ? OSOT_WAVES40 : 1))) osot_control_cycle_kernel(const DevUpdate U, const DevPlan P, const DevBatch D, const DevControl F) {
    OSOT_DYNAMIC_LDS(osot_smem);
    const long long inst0 = dispatch_instance(D, osot_smem);
    if (inst0 < 0) return;
    // (round 5) the producer WITH its collision-pair stage (closest points, distances, distance-Jacobian rows straight into the
    // CollisionAvoidance leaf buffers: velocity/CollisionAvoidance.cpp:96-152) in every instantiation that can meet inequality rows;
    // the stage is skipped at run time when the model has no pairs or the batch no pair outputs
    constexpr bool PAIRS = !BOX;
    // ROLLOUT (round 5, osot_control_rollout): the loop of the reference's example itself (coman_ik.cpp:174-219) for this robot,
    // `steps` times kinematics -> update -> cascade -> q += dq by the same wavefront.  As one launch per step every robot waits, at
    // every step, for the slowest robot of its sub-batch (the launch IS its longest job) and every step pays a dispatch; here a
    // robot's cycles follow each other directly and the launch ends with the robot whose SUM over the steps is largest -- which
    // the law of large numbers keeps near the mean.  The robots are independent, so the results are those of `steps` launches.
    const int steps = ROLL ? (F.steps > 1 ? F.steps : 1) : 1;
    int sticky = 0;
    for (int t = 0; t < steps; ++t) {
        // (ROLL: the instance number and the lane are made opaque once per cycle -- otherwise every address and mask derived from them
        //  is loop-invariant, hoisted out of the rollout loop and kept alive across the whole body: 22 .. 71 spilled registers)
        const long long inst = ROLL ? (long long)(((unsigned long long)(unsigned)launder_s(uniform_i((int)(inst0 >> 32))) << 32) | (unsigned)launder_s(uniform_i((int)inst0))) : inst0;
        const unsigned tid = ROLL ? (unsigned)launder_i((int)threadIdx.x) : threadIdx.x;
        if constexpr (NP == 32) {
            const int sub = (int)(tid >> 5), j = (int)(tid & 31u);
            kin_instance<PAIRS, 32>(F.K, F.Bt, inst, sub == 0, j, reinterpret_cast<double*>(osot_smem) + sub * kin_lds_doubles<32>(PAIRS));
        } else {
            kin_instance<PAIRS, 64>(F.K, F.Bt, inst, true, (int)tid, reinterpret_cast<double*>(osot_smem));
        }
        workgroup_fence();      // the producer's global stores (poses, rows of A_k, CoM) are visible to the update's loads (same workgroup)
        __syncthreads();
        update_body(OSOT_KERNARG_PTR(DevUpdate, U), inst, (int)tid, osot_smem);
        workgroup_fence();
        __syncthreads();
        cascade_body<NP, false, EXTRA, BOX>(P, D, inst, (int)tid, osot_smem);
        const int n = P.n, i = (int)tid;
        if (F.q_int && i < n) F.q_int[inst * n + i] += D.dq[inst * n + i];       // q += dq (the lane that stored dq[i] reads it back: its own store)
        if (ROLL && F.dq_steps && i < n) F.dq_steps[((long long)t * D.B + inst) * n + i] = D.dq[inst * n + i];
        if (ROLL && (steps > 1 || F.status_steps)) {
            workgroup_fence();      // q, dq and the status of this cycle before the next one's loads
            __syncthreads();
            const int st = uniform_i(D.status[inst]);
            if (F.status_steps && i == 0) F.status_steps[(long long)t * D.B + inst] = st;
            if (sticky == 0) sticky = st;
        }
    }
    if (ROLL && steps > 1 && threadIdx.x == 0) D.status[inst0] = sticky;
}

}  // namespace osot

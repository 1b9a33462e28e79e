// substep_hooke_stage.inc.hpp -- gravity + Hooke + the Verlet arithmetic of substep `it` (cloth.pyx:216-256) into REGISTERS, run while the
// strain sweep of substep it - 1 may still be walking (round 6: the substep loop is rotated, episode_loop.hpp).
// A FRAGMENT of k_run_schedule (episode_loop.hpp), included at its place in the kernel body: not a function (see substep_collision.inc.hpp for why).
// Names it uses from the kernel body: pm, Ak_, tid, lane, it, sc, dz_up / dxp / dyp / dzp, cur, misc, smem, lay, pvx/pvy/pvz, gt, rr, vm, lean_entry, lean_rest,
// rest_at; it defines nx / ny / nz (the new positions of the owned particles, committed by substep_write.inc.hpp behind the barrier).
//
// Nothing is written here -- not the positions, not the owner's previous positions, not the gripper's adjust -- so the stage is SPECULATIVE: if the
// loop ends at the barrier behind it (tear, time slice) its results are simply dropped.
//   * frontier: the sweeping wave publishes in misc[MISC_FRONT] the first window it has NOT finished (INT_MAX: no sweep in flight). A particle's
//     position is final for the substep once every window that holds one of its springs is finished; its Hooke sum reads the particle and its
//     twelve neighbours, so it may start when the frontier has passed ready[i] = the last window touching any of the thirteen (static per grid,
//     StepArgs::ready, one table in L2 for all cloths). The wait is per wave and per particle slot (the wave's largest ready[]); the sweeping wave itself gets here when it is done.
//   * Gripper.adjust of THIS substep (gripper.pyx:55-66, called before update(): cloth_env.py:358-363) has not been applied to LDS yet (the sweep
//     wave writes pinned records back unchanged: a concurrent adjust would be lost); a grabbed NEIGHBOUR is therefore read as delta + position,
//     the same IEEE operation the adjust performs. Grabbed particles are few: the hot loop only accumulates "did I meet one" (one instruction per
//     spring) and a particle that did recomputes its sum on a cold path with the adjusted neighbours. The particle's own adjust does not matter
//     here: a grabbed particle is pinned, and Verlet skips it (cloth.pyx:244).
        T nx[PPT], ny[PPT], nz[PPT];
#pragma unroll
        for (int q = 0; q < PPT; q++) nx[q] = ny[q] = nz[q] = (T)0;
        if ((pm & PH_HOOKE) && it < n_total_) {
            // (its own copy of the kernel-argument pointer: the phase's opaque redefinition must not leave this conditional region as a phi)
            KArgsC<T> *Ak_ = (KArgsC<T> *)__builtin_amdgcn_kernarg_segment_ptr();
            CLOTH_PHASE_ARGS()
            // adjust of substep `it`: lift (0, 0, dz_up), pull (dx, dy, dz), else none (cloth_env.py:352-367)
            const bool adj_up = it < sc.n_up_end, adj_pull = it >= sc.n_uprest_end && it < sc.n_pull_end;
            const uint32_t madj = (adj_up || adj_pull) ? (uint32_t)CNT_GRAB_MASK : 0u;
            const T aax = adj_pull ? dxp : (T)0, aay = adj_pull ? dyp : (T)0, aaz = adj_up ? dz_up : (adj_pull ? dzp : (T)0);
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                if constexpr (PIPE) {
                    // wait until the sweep of the previous substep has passed every window that touches this slot's particles or their neighbours
                    // (no sweep in flight -- the frontier at INT_MAX --: the table is not even read)
                    int f_ = __builtin_amdgcn_readfirstlane(*(volatile int *)&misc[MISC_FRONT]);
                    if (f_ != 0x7fffffff) {
                        const int mine_ = tid + q * NT < P ? (int)Ak_->ready[tid + q * NT] : -1;
                        int need_ = -__builtin_amdgcn_readlane(wave_incl_min(-mine_), 63);
#ifndef CLOTHHIP_PIPE_NO_SIMD_DEFER
                        // the wave that shares the sweeping wave's SIMD (a workgroup's waves go round the four SIMDs: w and w + 4 meet) stays out of its
                        // way: every instruction it issued there would come straight out of the sweep's issue slots (measured: the walk 20 % slower)
                        if (NT >= 512 && (tid >> 6) == SW - 4) need_ = 0x7ffffffe;
#endif
                        while (f_ <= need_) {
#ifndef CLOTHHIP_PIPE_SLEEP
#define CLOTHHIP_PIPE_SLEEP 2
#endif
                            __builtin_amdgcn_s_sleep(CLOTHHIP_PIPE_SLEEP);
                            f_ = __builtin_amdgcn_readfirstlane(*(volatile int *)&misc[MISC_FRONT]);
                        }
                    }
                    asm volatile("" ::: "memory");
                }
                // a real branch per particle: each particle's 12 springs form their own scheduling region, which
                // keeps the register allocator from interleaving all PPT*12 spring evaluations at once
                if (tid + q * NT < P) {
                    const Pt<T> me = cur[tid + q * NT];
                    T fx = (T)0 + (T)0, fy = (T)0 + (T)0, fz = (T)0 + k.mg;
                    uint32_t gl[HK_SLOTS];
                    uint32_t gacc = 0u;                                     // grab counts of the neighbours met (adjust substeps only)
                    int iq_ = tid + q * NT; uint32_t vq_ = vm_of(q, iq_);
                    if (LEAN) asm volatile("" : "+v"(iq_), "+v"(vq_));     // opaque: the stencil is recomputed every substep, not hoisted and held
#pragma unroll
                    for (int sl = 0; sl < HK_SLOTS; sl++)
                        gl[sl] = LEAN ? lean_entry(iq_, vq_, sl) : (GT_REG ? gt[GT_REG ? q : 0][sl] : Ak_->gather[sl * Ppad + tid + q * NT]);
#if defined(CLOTHHIP_MUTATE) && CLOTHHIP_MUTATE == 1
                    // MUTANT 1 (tools/run_mutants.sh; never a product build): ONE particle adds two of its incident springs in swapped list
                    // order (stencil positions 2 and 3: both shearing springs, same rest-length class) -- cloth.pyx:221-237 keeps list order
                    if (iq_ == P / 2) {
                        const uint32_t t_ = gl[2]; gl[2] = gl[3]; gl[3] = t_;      // (rest lengths follow the entry's table slot; the register-held
                                                                                   //  ones of the 256 x 3 debug variant are one value per class on the flat tiers)
                    }
#endif
                    // software pipeline: the neighbour records of the next springs are in flight while spring sl is
                    // evaluated (left to itself the scheduler, which minimises live registers at this kernel's pressure, issues
                    // each 16-byte read right before its use and waits out the whole LDS latency 12 times per particle)
                    constexpr int HK_AHEAD = 2;
                    Pt<T> nbq[HK_AHEAD];
#pragma unroll
                    for (int sl = 0; sl < HK_AHEAD; sl++) {
                        uint32_t g = gl[sl];
                        asm volatile("" : "+v"(g));         // opaque: keeps the address math inside the substep loop
                        gl[sl] = g;
                        nbq[sl] = cur[g & HK_NBR_MASK];
                    }
#pragma unroll
                    for (int sl = 0; sl < HK_SLOTS; sl++) {
                        const uint32_t g = gl[sl];
                        const Pt<T> nb = nbq[sl % HK_AHEAD];
                        if (sl + HK_AHEAD < HK_SLOTS) {
                            uint32_t gn = gl[sl + HK_AHEAD];
                            asm volatile("" : "+v"(gn));
                            gl[sl + HK_AHEAD] = gn;
                            nbq[sl % HK_AHEAD] = cur[gn & HK_NBR_MASK];
                        }
                        __builtin_amdgcn_sched_barrier(0);  // the reads above stay above the arithmetic below
                        const T r = LEAN ? lean_rest(sl) : (REST_R ? rr[REST_R ? q : 0][sl] : rest_at((g >> HK_POS_SHIFT) & HK_POS_MASK));
                        const T kk = (LEAN ? lean_bend(sl) : (g & HK_BEND) != 0u) ? k.ks_bend : k.ks_str;
                        // (an absent slot reads the particle itself / particle 0: a grabbed one there only sends the wave down the cold path for nothing)
                        gacc |= w_cnt(nb.w) & madj;
                        const T dx = nb.x - me.x, dy = nb.y - me.y, dz = nb.z - me.z;
                        const T l = fastnorm<T>(dx, dy, dz);                                      // :231
                        const T fm = dev_div<T>(kk * (l - r), l);                                 // :232
                        const bool valid = (g & HK_VALID) != 0u;
                        fx = valid ? mad<T>(fm, dx, fx) : fx; fy = valid ? mad<T>(fm, dy, fy) : fy; fz = valid ? mad<T>(fm, dz, fz) : fz;   // :236-237
                    }
#ifdef CLOTHHIP_EXP_NOCOLD
                    if (false) {
#else
                    if (__builtin_expect(__any(gacc != 0u), 0)) {
#endif
                        // cold path (a wave that holds a neighbour of a grabbed particle, lift / pull substeps only): the sum again, every grabbed
                        // neighbour moved by the gripper first -- p <- x; x <- delta + x, once per entry in grabbed_pts (gripper.pyx:60-66)
                        fx = (T)0 + (T)0; fy = (T)0 + (T)0; fz = (T)0 + k.mg;
#pragma unroll 1
                        for (int sl = 0; sl < HK_SLOTS; sl++) {
                            uint32_t g = LEAN ? lean_entry(iq_, vq_, sl) : (GT_REG ? gt[GT_REG ? q : 0][0] : Ak_->gather[sl * Ppad + iq_]);
                            if constexpr (GT_REG) {         // (a register array cannot be indexed by a loop counter without going to scratch)
#pragma unroll
                                for (int s2 = 1; s2 < HK_SLOTS; s2++) g = sl == s2 ? gt[GT_REG ? q : 0][s2] : g;
                            }
                            Pt<T> nb = cur[g & HK_NBR_MASK];
                            const int m_ = (int)(w_cnt(nb.w) & madj);
                            for (int r_ = 0; r_ < m_; r_++) { nb.x = aax + nb.x; nb.y = aay + nb.y; nb.z = aaz + nb.z; }
                            T r = LEAN ? lean_rest(sl) : rest_at((g >> HK_POS_SHIFT) & HK_POS_MASK);
                            if constexpr (REST_R) {
#pragma unroll
                                for (int s2 = 0; s2 < HK_SLOTS; s2++) r = sl == s2 ? rr[REST_R ? q : 0][s2] : r;
                            }
                            const T kk = (LEAN ? lean_bend(sl) : (g & HK_BEND) != 0u) ? k.ks_bend : k.ks_str;
                            const T dx = nb.x - me.x, dy = nb.y - me.y, dz = nb.z - me.z;
                            const T l = fastnorm<T>(dx, dy, dz);
                            const T fm = dev_div<T>(kk * (l - r), l);
                            const bool valid = (g & HK_VALID) != 0u;
                            fx = valid ? mad<T>(fm, dx, fx) : fx; fy = valid ? mad<T>(fm, dy, fy) : fy; fz = valid ? mad<T>(fm, dz, fz) : fz;
                        }
                    }
                    nx[q] = mad<T>(fx, k.dsm, mad<T>(k.damp, me.x - pvx[q], me.x));               // :249
                    ny[q] = mad<T>(fy, k.dsm, mad<T>(k.damp, me.y - pvy[q], me.y));
                    nz[q] = mad<T>(fz, k.dsm, mad<T>(k.damp, me.z - pvz[q], me.z));
                }
            }
        }

// substep_strain.inc.hpp -- strain limit + tear (cloth.pyx:258-296): parallel pre-pass, then the ordered sweep over the window table (phase_strain.hpp); RELAXED builds: coloured order
// A FRAGMENT of k_run_schedule (episode_loop.hpp), included at its place in the kernel body: not a function. Turning the substep's phases into
// __forceinline__ functions over a context struct was tried (round 5): same instructions, but the register allocation of the 128-VGPR variants
// shifts -- three more scratch reloads in the substep loop, -1.4 % on the headline -- so the split is textual and the ISA is bit-identical to the
// one-file kernel's. Names it uses from the kernel body: pm, Ak_, tid, lane, cur, misc, wtab, g_rest, pslot, gt, rr, vm, rc, lean_entry, lean_rest, rest_at, st_*, tph / TSTAMP, mode (census build).
        // ---- strain limit + tear (cloth.pyx:258-296) ---------------------------------------------------
        // (1) all threads: which springs would stretch/tear at the CURRENT positions? Only the first and the last of them
        //     (in window-table order) are kept: a spring untouched by earlier corrections of the sweep behaves exactly as
        //     evaluated here, so nothing before the first needs a look, and nothing behind the last unless a correction
        //     reaches it. No spring flagged: the sweep is skipped (a cloth at rest).
        // (2) wave 0 walks the windows in between (strain_sweep above).
        if constexpr (RELAXED) {
            // Coloured order (NOT the reference's list order, cloth.pyx:258-296): the six springs a particle owns (to r-1, c-1, the two
            // diagonals, r-2, c-2) in two parity classes each -- twelve classes whose springs share no particle --, one class after the
            // other, every class in parallel by the owners of its springs. Same test, same correction per spring.
            CLOTH_PHASE_ARGS()
            static_assert(!RELAXED || LEAN, "the relaxed-order companion exists for the LEAN arithmetic (stencil from the grid position, palette rest lengths)");
            int tear_ = 0;
#pragma unroll 1
            for (int col = 0; col < 12; col++) {
                const int kind = col >> 1, par = col & 1;
#pragma unroll
                for (int q = 0; q < PPT; q++) {
                    const int i = tid + q * NT;
                    const int r_ = (int)(rc[RELAXED ? q : 0] & 0xFFu), c_ = (int)(rc[RELAXED ? q : 0] >> 8);
                    const int key = (kind == 1 || kind == 5) ? c_ : r_;
                    const bool on = i < P && ((vm[(LEAN && !LEAN64) ? q : 0] >> kind) & 1u) && (((kind >= 4 ? key >> 1 : key) & 1) == par);
                    if (on) {
                        const int j = i + (kind == 0 ? -Ak_->N : kind == 1 ? -1 : kind == 2 ? -Ak_->N - 1 : kind == 3 ? -Ak_->N + 1 : kind == 4 ? -2 * Ak_->N : -2);
                        const Pt<T> a_ = cur[j], b_ = cur[i];                                  // ptA (the earlier point), ptB (the owner)
                        const uint32_t ca = w_cnt(a_.w), cb = w_cnt(b_.w);
                        const T rest = kind >= 4 ? Ak_->pal_bend : (kind >= 2 ? Ak_->pal_shear : Ak_->pal_struct);
                        const T dx = a_.x - b_.x, dy = a_.y - b_.y, dz = a_.z - b_.z;
                        const T len = fastnorm<T>(dx, dy, dz);
                        if (!((ca != 0) & (cb != 0))) {
                            if (len > rest * k.tear_thresh) tear_ = 1;
                            const T t11 = rest * k.c11;
                            if (len > t11) {
                                const T ux = dev_div<T>(dx, len), uy = dev_div<T>(dy, len), uz = dev_div<T>(dz, len);
                                const T extra = len - t11;
                                const T wa = ca != 0 ? (T)0 : (cb != 0 ? (T)1 : (T)0.5), wb = cb != 0 ? (T)0 : (ca != 0 ? (T)1 : (T)0.5);
                                const T ea = extra * wa, eb = extra * wb;
                                if (ca == 0) cur[j] = Pt<T>{mad<T>(-ux, ea, a_.x), mad<T>(-uy, ea, a_.y), mad<T>(-uz, ea, a_.z), a_.w};
                                if (cb == 0) cur[i] = Pt<T>{mad<T>(ux, eb, b_.x), mad<T>(uy, eb, b_.y), mad<T>(uz, eb, b_.z), b_.w};
                            }
                        }
                    }
                }
                __syncthreads();
            }
            if (__any(tear_) && lane == 0) misc[0] = 1;
            __syncthreads();
        } else
        if (pm & PH_STRAIN) {
            CLOTH_PHASE_ARGS()
            {
                // Every spring is tested once, by the owner of its ptB (the particle the reference appended it for):
                // the owner holds the spring's gather entry (neighbour = ptA, table slot) and, with
                // REST_REG, its rest length in registers, so the pre-pass needs one 16-byte LDS read per spring.
                int nact = 0, pmin = 0x7fffffff, pmax = -1;     // flagged springs; the first / last of them in table order
#pragma unroll
                for (int q = 0; q < PPT; q++) {
                    if (tid + q * NT < P) {
                        const Pt<T> me = cur[tid + q * NT];
                        const uint32_t cme_ = w_cnt(me.w);
                        uint32_t gl[HK_SLOTS / 2];
                        int iq_ = tid + q * NT; uint4 lw_ = uint4{0u, 0u, 0u, 0u}; if constexpr (LEAN64) lw_ = lstc[iq_]; uint32_t vq_ = LEAN64 ? lw_.x : vm[(LEAN && !LEAN64) ? q : 0];
                        if (LEAN) asm volatile("" : "+v"(iq_), "+v"(vq_));
#pragma unroll
                        for (int sl = 0; sl < HK_SLOTS / 2; sl++)
                            gl[sl] = LEAN ? lean_entry(iq_, vq_, sl) : (GT_REG ? gt[GT_REG ? q : 0][sl] : Ak_->gather[sl * Ppad + tid + q * NT]);
                        // software pipeline, as in the Hooke phase: two neighbour reads in flight ahead of the test
                        constexpr int PP_AHEAD = 2;
                        Pt<T> nbq[PP_AHEAD];
#pragma unroll
                        for (int sl = 0; sl < PP_AHEAD; sl++) {
                            uint32_t g = gl[sl];
                            asm volatile("" : "+v"(g));
                            gl[sl] = g;
                            nbq[sl] = cur[g & HK_NBR_MASK];
                        }
                        // (a) branch-free: which of the six springs come within the slack band of their limit at all?
                        uint32_t cand = 0u;
                        T l2s[HK_SLOTS / 2];
#pragma unroll
                        for (int sl = 0; sl < HK_SLOTS / 2; sl++) {       // own springs come first in ascending list order
                            const uint32_t g = gl[sl];
                            const Pt<T> nb = nbq[sl % PP_AHEAD];
                            if (sl + PP_AHEAD < HK_SLOTS / 2) {
                                uint32_t gn = gl[sl + PP_AHEAD];
                                asm volatile("" : "+v"(gn));
                                gl[sl + PP_AHEAD] = gn;
                                nbq[sl % PP_AHEAD] = cur[gn & HK_NBR_MASK];
                            }
                            __builtin_amdgcn_sched_barrier(0);
                            T r = LEAN64 ? lean_rest64(sl, lw_) : LEAN ? lean_rest(sl) : (REST_R ? rr[REST_R ? q : 0][sl] : rest_at((g >> HK_POS_SHIFT) & HK_POS_MASK));
                            asm volatile("" : "+v"(r));     // or the thresholds below are hoisted out of the substep loop
                                                            // for all 18 springs and live in scratch
                            const T dx = nb.x - me.x, dy = nb.y - me.y, dz = nb.z - me.z;   // (ptA - ptB), as :270
                            const T len2 = sumsq<T>(dx, dy, dz);
                            l2s[sl] = len2;
                            const T t11 = r * k.c11, tt = r * k.tear_thresh;
                            const T tmin = t11 < tt ? t11 : tt;
#if defined(CLOTHHIP_MUTATE) && CLOTHHIP_MUTATE == 4       // MUTANT 4 (see strain_sweep_lean): the pre-pass flags both-pinned springs too
                            const bool pre = ((g & (HK_VALID | HK_ASB)) == (HK_VALID | HK_ASB)) &
                                             (len2 > tmin * tmin * ((T)1 - filt_slack<T>()));
#else
                            const bool pre = ((g & (HK_VALID | HK_ASB)) == (HK_VALID | HK_ASB)) &
                                             !((cme_ != 0) & (w_cnt(nb.w) != 0)) &
                                             (len2 > tmin * tmin * ((T)1 - filt_slack<T>()));
#endif
                            cand |= pre ? (1u << sl) : 0u;
                        }
                        // (b) those few: inside the slack band around the limit the sweep's exact test (:270-275) decides: a
                        // spring that sits exactly ON its limit (left there by an earlier substep's correction) is then not
                        // flagged, and a cloth at rest skips the sweep altogether
                        if (cand) {
#pragma unroll
                            for (int sl = 0; sl < HK_SLOTS / 2; sl++) {
                                if (cand & (1u << sl)) {
                                    // (LEAN: the spring's table slot is read from the gather table only now that it is needed: the table
                                    //  is compacted, the sl-th stencil position is the particle's popcount(valid below sl)-th entry)
                                    const uint32_t pos_ = TAB == 2 ? (uint32_t)pslot[sl * Ppad + iq_] : ((LEAN ? Ak_->gather[__popc(vq_ & ((1u << sl) - 1u)) * Ppad + iq_] : gl[sl])      // (the opaque copies: nothing of this is hoisted out of the substep loop and held)
                                                           >> HK_POS_SHIFT) & HK_POS_MASK;
                                    T r = LEAN64 ? lean_rest64(sl, lw_) : LEAN ? lean_rest(sl) : (REST_R ? rr[REST_R ? q : 0][sl] : rest_at(pos_));
                                    asm volatile("" : "+v"(r));
                                    const T len2 = l2s[sl];
                                    const T t11 = r * k.c11, tt = r * k.tear_thresh;
                                    const T tmin = t11 < tt ? t11 : tt;
                                    bool flag = len2 > tmin * tmin * ((T)1 + filt_slack<T>());
                                    if (!flag) { const T len = dev_sqrt<T>(len2); flag = len > t11 || len > tt; }
                                    if (flag) { nact++; pmin = (int)pos_ < pmin ? (int)pos_ : pmin; pmax = (int)pos_ > pmax ? (int)pos_ : pmax; }
#ifdef CLOTHHIP_CELL_COUNTERS
                                    if (flag) atomicOr(&misc[13 + (((int)pos_ >> 6) >> 5 & 1)], 1 << (((int)pos_ >> 6) & 31));
#endif
                                }
                            }
                        }
                    }
                }
                if (__any(nact)) {
                    const int min_ = __builtin_amdgcn_readlane(wave_incl_min(pmin), 63);         // DPP: no LDS round trips
                    const int max_ = -__builtin_amdgcn_readlane(wave_incl_min(-pmax), 63);
                    if (lane == 0) { misc[1] = 1; atomicMin(&misc[10], min_); atomicMax(&misc[11], max_); }
                }
            }
            __syncthreads();
            TSTAMP(8)
#ifdef CLOTHHIP_CELL_COUNTERS
            int swept_ = 0;
#endif
            if (SWEEP_MW) {
                // every wave of the cloth takes part (strain_sweep_mw): the decision and the walk's bounds are read by all of them
                // before the sweep's first barrier and reset by wave 0 behind its last
                if (misc[1] || (pm & PH_NOSKIP)) {
                    const bool all_ = (pm & PH_NOSKIP) != 0;
                    const int w0 = __builtin_amdgcn_readfirstlane(all_ ? 0 : (misc[10] >> 6));
                    const int w1 = __builtin_amdgcn_readfirstlane(all_ ? KA_NW(Ak_) - 1 : (misc[11] >> 6));
                    const bool tic = __builtin_amdgcn_readfirstlane(!(k.tear_thresh < k.c11) ? 1 : 0) != 0;
                    const int wave_ = __builtin_amdgcn_readfirstlane(tid >> 6);
                    const int wl_ = (Ak_->Spad >> 6) - 1;                   // the table's last (padding, empty) window
                    int *const sw_ = misc + 24, *const st_ = misc + 20;
#ifdef CLOTHHIP_MW_PRIO
                    __builtin_amdgcn_s_setprio(CLOTHHIP_MW_PRIO);
#endif
                    const int tear = tic ? strain_sweep_mw<T, v_ldstab(TAB), NT / 64, SWEEP_STATS, true>(cur, wtab, Ak_->wt_ent, g_rest, Ak_->wt_dep, w0, w1, wl_, KA_RSHIFT(Ak_), k, lane, wave_, sw_, st_)
                                         : strain_sweep_mw<T, v_ldstab(TAB), NT / 64, SWEEP_STATS, false>(cur, wtab, Ak_->wt_ent, g_rest, Ak_->wt_dep, w0, w1, wl_, KA_RSHIFT(Ak_), k, lane, wave_, sw_, st_);
#ifdef CLOTHHIP_MW_PRIO
                    __builtin_amdgcn_s_setprio(0);
#endif
                    if (__any(tear) && lane == 0) misc[0] = 1;
#ifdef CLOTHHIP_CELL_COUNTERS
                    swept_ = 1;
#endif
                    if (tid == 0) {
                        misc[15]++; misc[1] = 0; misc[10] = 0x7fffffff; misc[11] = -1;
                        if (SWEEP_STATS) {
                            st_passes += st_[0]; st_commits += st_[1]; st_windows += st_[2];
#ifdef CLOTHHIP_MW_ROUNDS
                            st_commits += st_[3] - st_[1]; st_[3] = 0;
#endif
                            st_[0] = 0; st_[1] = 0; st_[2] = 0;
                        }
                    }
                }
            } else
            if (tid < 64 && (misc[1] || (pm & PH_NOSKIP))) {
                __builtin_amdgcn_s_setprio(3);            // the serial sweep is the critical path of the whole cloth
                const bool all_ = (pm & PH_NOSKIP) != 0;
                const int w0 = __builtin_amdgcn_readfirstlane(all_ ? 0 : (misc[10] >> 6));
                const int w1 = __builtin_amdgcn_readfirstlane(all_ ? KA_NW(Ak_) - 1 : (misc[11] >> 6));
                if (lane == 0) misc[15]++;               // sweeps run (clothhip_debug_stats): in LDS -- as a register it was spilled, reloaded and stored by every sweep
                // tear_thresh >= 1.1 (every shipped configuration): only a stretching spring can tear, the test sits in the commit
                const bool tic = __builtin_amdgcn_readfirstlane(!(k.tear_thresh < k.c11) ? 1 : 0) != 0;
#ifdef CLOTHHIP_CELL_COUNTERS
                const unsigned long long fmask_ = (unsigned long long)(uint32_t)misc[13] | ((unsigned long long)(uint32_t)misc[14] << 32);
                swept_ = 1;
#else
                const unsigned long long fmask_ = 0ull;
#endif
                const int tear = tic ? (SWEEP_LEAN ? strain_sweep_lean<T, v_ldstab(TAB), SWEEP_STATS, SWEEP_AHEAD>(cur, wtab, Ak_->wt_ent, g_rest, Ak_->wt_dep, w0, w1, KA_RSHIFT(Ak_), k,
                                                                                                  lane, st_windows, st_passes, st_commits)
                                                   : strain_sweep<T, v_ldstab(TAB), SWEEP_TIMED, SWEEP_STATS, true>(cur, wtab, Ak_->wt_ent, g_rest, Ak_->wt_dep, w0, w1, Ak_->nW, KA_RSHIFT(Ak_), k,
                                                                                                 lane, st_windows, st_passes, st_commits, tph, fmask_))
                                     : strain_sweep<T, v_ldstab(TAB), SWEEP_TIMED, SWEEP_STATS, false>(cur, wtab, Ak_->wt_ent, g_rest, Ak_->wt_dep, w0, w1, Ak_->nW, KA_RSHIFT(Ak_), k,
                                                                                                  lane, st_windows, st_passes, st_commits, tph, fmask_);
                if (__any(tear) && lane == 0) misc[0] = 1;
                if (lane == 0) { misc[1] = 0; misc[10] = 0x7fffffff; misc[11] = -1; }
#ifdef CLOTHHIP_CELL_COUNTERS
                if (lane == 0) { misc[13] = 0; misc[14] = 0; }
#endif
                __builtin_amdgcn_s_setprio(0);
            }
            __syncthreads();
#ifndef CLOTHHIP_SWEEP_STAMPS          // (that build uses slots 9-11 for the sweep's passes)
            TSTAMP(9)
#endif
#ifdef CLOTHHIP_CELL_COUNTERS
            // census (wave 0): a substep in which nothing was adjusted, the plane restored every unpinned particle to its old
            // position (friction 1) and no spring was over-stretched leaves the positions as they were: tph[1] counts those,
            // tph[2] those whose predecessor was one too (state(t+1) == state(t): a fixed point)
            if (tid < 64) {
                const bool frozen_ = mode != 1 && misc[12] == 0 && !swept_ && k.one_m_fric == (T)0;
                tph[1] += frozen_ ? 64 : 0; tph[2] += misc[16] == 0 ? 64 : 0;     // [2]: no particle changed its collision cell in this substep
                frozen_prev_ = frozen_;
            }
            __syncthreads();
            if (tid == 0) { misc[12] = 0; misc[16] = 0; }
#endif
        }

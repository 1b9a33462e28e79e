// episode_finish.inc.hpp -- after a run of substeps (fused launches): time-slice parking, in-kernel metrics, the step / reset records, terminal test (cloth_env.py:536-715, :780-789)
// A FRAGMENT of k_run_schedule (episode_loop.hpp), included at its place in the kernel body: not a function. Turning the substep's phases into
// __forceinline__ functions over a context struct was tried (round 5): same instructions, but the register allocation of the 128-VGPR variants
// shifts -- three more scratch reloads in the substep loop, -1.4 % on the headline -- so the split is textual and the ISA is bit-identical to the
// one-file kernel's. Names it uses from the kernel body: Fp, eps, sc, it_next, done, resumed_run, resume_done, tid, cur, misc, P, smem, lay, init_lds, account, OP_*.
        // ---- after the run: everything is re-read from LDS -----------------------------------------------------------
        {
            const FusedArgs<T> &F = *Fp;
            __syncthreads();
            if (it_next >= 0) {                          // cut by the time slice: park the run and leave
                if (tid == 0) {
                    eps->done_total += done - (resumed_run ? resume_done : 0);
                    account(eps->op, done - (resumed_run ? resume_done : 0));
                    EpResume *rs_ = F.resume + e;
                    rs_->valid = 1; rs_->it = it_next; rs_->done_partial = done; rs_->sc = sc; rs_->eps = *eps;
                    if (eps->rp >= 0 && F.resets != nullptr) {
                        ClothResetRecord *rr_ = F.resets + ((size_t)e * F.n_scripts + eps->n_resets);
                        rs_->rr = *rr_; rr_->consumed = 2;
                    }
                }
                __syncthreads();
                break;
            }
            const int tear_now = __builtin_amdgcn_readfirstlane(misc[0]);
            const int op = eps->op, rp = eps->rp, t_slot = eps->t_slot, n_resets = eps->n_resets;
            double mo[4] = {0.0, 0.0, 0.0, 0.0};
            if (op == OP_ACTION || op == OP_RESET_COND || op == OP_RESET_END) {
                // cloth_env.py:1020-1098 on the LDS-resident state; the sort buffers borrow the LDS behind the particle records
                auto src = [&](int i, double &x, double &y, double &z) { const Pt<T> c = cur[i]; x = (double)c.x; y = (double)c.y; z = (double)c.z; };
                // (the sort buffers and the hull stack live BEHIND the window table -- hash table, member lists, cell-ordered copy: all
                //  rebuilt below --, so the table itself stays in LDS for the whole launch and is not re-read from L2 after every action)
                metrics_block<NT, T, decltype(src), v_hull_idx(TAB, (int)sizeof(T), NT, PPT)>(src, P, F.NS, F.NH, smem + lay.hkey, tid, F.half_thickness, mo);
                init_lds(tear_now, nullptr, nullptr);
                __syncthreads();
            }
            if (op == OP_ACTION && F.obs) {                                                       // '1d' observation, cloth_env.py:196-200
                float *o_ = F.obs + ((size_t)t_slot * F.E + e) * 3 * P;
                for (int i = tid; i < P; i += NT) { const Pt<T> c = cur[i]; o_[3 * i] = (float)c.x; o_[3 * i + 1] = (float)c.y; o_[3 * i + 2] = (float)c.z; }
            }
            if (op == OP_RESET_END && F.reset_obs) {                                              // what env.reset() returns
                float *o_ = F.reset_obs + ((size_t)e * F.n_scripts + n_resets) * 3 * P;
                for (int i = tid; i < P; i += NT) { const Pt<T> c = cur[i]; o_[3 * i] = (float)c.x; o_[3 * i + 1] = (float)c.y; o_[3 * i + 2] = (float)c.z; }
            }
            if (tid == 0) {
                if (F.budget_ticks != 0 && __builtin_amdgcn_s_memrealtime() - eps->t_launch >= F.budget_ticks) eps->stop = 1;
                eps->done_total += done - (resumed_run ? resume_done : 0);
                account(op, done - (resumed_run ? resume_done : 0));
                if (op == OP_ACTION) {
                    const int ep_steps = eps->ep_steps + 1;
                    const bool oob_ = mo[2] != 0.0;
                    // _terminal (cloth_env.py:684-715)
                    const bool dn = ep_steps >= F.ep.max_actions || tear_now != 0 || oob_ || mo[0] > F.ep.coverage_done;
                    ClothStepRecord *r_ = F.records + ((size_t)t_slot * F.E + e);
                    r_->action[0] = eps->act[0]; r_->action[1] = eps->act[1]; r_->action[2] = eps->act[2]; r_->action[3] = eps->act[3];
                    r_->coverage = mo[0]; r_->variance_inv = mo[1]; eps->last_cov = mo[0];
                    r_->executed = done; r_->n_grabbed = eps->n_grab; r_->iters_pull = eps->iters_pull;
                    r_->n_below_half_thickness = (int32_t)mo[3];
                    r_->ran = eps->decode_err ? 2 : 1; r_->oob = oob_ ? 1 : 0; r_->tear = tear_now ? 1 : 0; r_->done = dn ? 1 : 0;
                    r_->reset_before = (uint8_t)eps->reset_mark;
                    eps->reset_mark = 0; eps->ep_steps = ep_steps; eps->ep_done = dn ? 1 : 0; eps->t_slot = t_slot + 1; eps->n_ran += 1;
                } else {
                    const bool rngm = F.mt != nullptr;
                    const ClothResetScript *scr = rngm ? nullptr : F.scripts + ((size_t)e * F.n_scripts + n_resets);
                    ClothResetRecord *rr_ = F.resets ? F.resets + ((size_t)e * F.n_scripts + n_resets) : nullptr;
                    if (op == OP_RESET_COND) {
                        const double cmin = rngm ? 0.90 : scr->pull[rp >> 1].coverage_min;
                        eps->rp = mo[0] >= cmin ? rp + 1 : 6;                                     // cloth_env.py:866
                    } else if (op == OP_RESET_PULL) {
                        const int p_ = rp >> 1;
                        if (rr_) {
                            rr_->executed[p_] = done; rr_->pulls_run = eps->rs_pulls + 1;
                            rr_->action[p_][0] = eps->act[0]; rr_->action[p_][1] = eps->act[1];
                            rr_->action[p_][2] = eps->act[2]; rr_->action[p_][3] = eps->act[3];
                        }
                        eps->rs_pulls += 1; eps->rp = rp + 1;
                    } else if (op == OP_RESET_SETTLE) {
                        if (rr_) rr_->settle_executed += done;
                        eps->rp = (with_tier2 && rp == 8) ? 0 : 7;
                    } else {                                                                      // OP_RESET_END
                        if (rr_) { rr_->start_coverage = mo[0]; rr_->start_variance_inv = mo[1]; rr_->tear = tear_now; }
                        eps->last_cov = mo[0];
                        // a conditional pull that ran consumed RNG draws the later scripts were drawn without (clothhip.h)
                        if (!rngm) {
                            int n_uncond = 0;
                            for (int p_ = 0; p_ < scr->n_pulls; p_++) n_uncond += scr->pull[p_].need_coverage ? 0 : 1;
                            if (eps->rs_pulls > n_uncond) eps->chain_ok = 0;
                        }
                        eps->n_resets = n_resets + 1; eps->reset_mark = n_resets + 1; eps->rp = -1;
                    }
                }
            }
            __syncthreads();
            if (op == OP_RESET_END && F.mt != nullptr && F.domrand_words != 0)                    // cloth_env.py:786-789
                mt_skip_block<NT>(F.mt + (size_t)e * MT_WORDS, F.domrand_words, tid);
        }

// episode_plan.inc.hpp -- the episode state machine's planning step (fused launches): which operation comes next for this cloth -- action (decode, Gripper.grab_top), reset stage, settle --, cloth_env.py:369-534, :717-987
// A FRAGMENT of k_run_schedule (episode_loop.hpp), included at its place in the kernel body: not a function. Turning the substep's phases into
// __forceinline__ functions over a context struct was tried (round 5): same instructions, but the register allocation of the 128-VGPR variants
// shifts -- three more scratch reloads in the substep loop, -1.4 % on the headline -- so the split is textual and the ISA is bit-identical to the
// one-file kernel's. Names it uses from the kernel body: Fp, eps, sc, do_run, tid, lane, cur, misc, P, Ppad, A, pvx/pvy/pvz, init_lds, OP_*.
            // ---- plan the next operation. Every thread evaluates the same transitions on the same LDS-resident state.
            const FusedArgs<T> &F = *Fp;
            int t_slot = eps->t_slot, rp = eps->rp;
            const int n_resets = eps->n_resets;
            // rp >= 0: the script of the reset in progress; else the env's next one, valid only while the chain is intact
            const bool have_scr = F.scripts != nullptr && n_resets < F.n_scripts && (rp >= 0 || eps->chain_ok);
            const ClothResetScript *scr = have_scr ? F.scripts + ((size_t)e * F.n_scripts + n_resets) : nullptr;
            // device-RNG resets (F.mt): the script is not read from a table but drawn from the env's numpy stream as the reset
            // proceeds, in the reference's order; its shape depends on the tier only
            uint32_t *const mt = F.mt ? F.mt + (size_t)e * MT_WORDS : nullptr;
            const bool rngm = mt != nullptr;
            const int tier = with_tier2 ? F.rng_tier : (F.rng_tier == 3 ? 3 : 1);
            auto s_n_pulls = [&]() { return rngm ? (tier == 1 ? 3 : (tier == 2 ? 2 : 1)) : scr->n_pulls; };
            auto s_settle = [&]() { return rngm ? (tier == 3 ? 800 : (tier == 2 ? 500 : 0)) : scr->settle_after; };
            auto s_need_cov = [&](int p_) { return rngm ? (tier == 1 && p_ == 2) : ((scr->pull[p_].need_coverage & 1) != 0); };
            int op = OP_ACTION;
            bool do_decode = false;
            double act[4] = {0.0, 0.0, 0.0, 0.0};
            double run_iters_up = F.ep.iters_up;
            do_run = false;
            if (rp < 0) {
                if (t_slot >= F.nT) break;
                // time slice: envs advance at their own pace, so a launch ends when its time budget is used up rather than when
                // the slowest env has finished a fixed number of actions. Decided by thread 0 between operations (also between a
                // reset and the first action of the new episode: the reset record tells the host). Which launch executes an
                // action never changes its result.
                if (eps->stop) break;
                if (eps->ep_done) {
                    __syncthreads();                     // everyone has read the state
                    if (rngm ? (n_resets < F.n_scripts) : (scr != nullptr && scr->valid)) {
                        // the Cloth(...) rebuild of ClothEnv.reset (cloth_env.py:737-746): nothing pinned, no tear
                        int side_ = 0;
                        bool t2_ = false;
                        if constexpr (with_tier2) t2_ = rngm && tier == 2;
                        if constexpr (with_tier2) if (t2_) {
                            // tier 2 (cloth.pyx:94-116): a vertical sheet at x = |noise| (init_side) or 1 - |noise|, one rand()
                            // per point in r-major order (row 0 draws too, its noise is zeroed), and rest lengths measured on
                            // these positions (cloth.pyx:417) -- in double, as the host's clothhip_init_grid does, through a
                            // scratch copy behind the particle records
                            double *dpos = reinterpret_cast<double *>(smem + lay.wtab);
                            if (tid == 0) {
                                side_ = mt_double(mt) > 0.5 ? 1 : 0;                             // cloth.pyx:75
                                const int N_ = KA_N(&A);
                                for (int r_ = 0; r_ < N_; r_++)
                                    for (int c_ = 0; c_ < N_; c_++) {
                                        double noise = mt_double(mt) * 0.01 - 0.005;             // :101
                                        if (r_ == 0) noise = 0;                                  // :102-103
                                        const int i = r_ * N_ + c_;
                                        dpos[3 * i] = side_ ? 0.0 + fabs(noise) : 1.0 - fabs(noise);   // :104-107
                                        dpos[3 * i + 1] = F.grid_dx * c_; dpos[3 * i + 2] = F.grid_dy * r_;   // :109-110
                                    }
                                eps->side = side_;
                            }
                            __syncthreads();
                            side_ = eps->side;
                            for (int i = tid; i < Ppad; i += NT)
                                cur[i] = i < P ? Pt<T>{(T)dpos[3 * i], (T)dpos[3 * i + 1], (T)dpos[3 * i + 2], w_make<T>(0u)}
                                               : Pt<T>{(T)0, (T)0, (T)0, w_make<T>(0u)};
#pragma unroll
                            for (int q = 0; q < PPT; q++) {
                                const int i = tid + q * NT;
                                if (i < P) { pvx[q] = (T)dpos[3 * i]; pvy[q] = (T)dpos[3 * i + 1]; pvz[q] = (T)dpos[3 * i + 2]; }
                            }
                            T *rw = F.rest_rw + (size_t)e * F.rest_stride;
                            for (int p_ = tid; p_ < KA_SPAD(&A); p_ += NT) {
                                const uint32_t en = F.wt_ent[p_];                                 // empty slots: ptA == ptB == 0 -> 0
                                const double *PA = dpos + 3 * (en & WT_IDX_MASK), *PB = dpos + 3 * ((en >> WT_IDX_BITS) & WT_IDX_MASK);
                                const double ux = PA[0] - PB[0], uy = PA[1] - PB[1], uz = PA[2] - PB[2];
                                rw[p_] = (T)sqrt(ux * ux + uy * uy + uz * uz);                    // cloth.pyx:417 via :17-18
                            }
                            __syncthreads();
                            init_lds(0, F.wt_ent, F.rest + (size_t)e * F.rest_stride);
                            if (REST_R) {
#pragma unroll
                                for (int q = 0; q < PPT; q++)
#pragma unroll
                                    for (int sl = 0; sl < HK_SLOTS; sl++) {
                                        const uint32_t g0 = GT_REG ? gt[GT_REG ? q : 0][sl] : 0u;
                                        rr[REST_R ? q : 0][sl] = rw[(g0 >> HK_POS_SHIFT) & HK_POS_MASK];
                                    }
                            }
                        }
                        if (!t2_) {
                            for (int i = tid; i < Ppad; i += NT)
                                cur[i] = Pt<T>{F.flat[i], F.flat[Ppad + i], F.flat[2 * Ppad + i], w_make<T>(0u)};
#pragma unroll
                            for (int q = 0; q < PPT; q++) {
                                const int i = tid + q * NT;
                                if (i < P) { pvx[q] = F.flat[i]; pvy[q] = F.flat[Ppad + i]; pvz[q] = F.flat[2 * Ppad + i]; }
                            }
                        }
                        if (tid == 0) {
                            misc[0] = 0;
                            eps->rp = t2_ ? 8 : 0; eps->rs_pulls = 0; eps->ep_steps = 0; eps->ep_done = 0;
                            if (rngm && !t2_) side_ = mt_double(mt) > 0.5 ? 1 : 0;               // cloth.pyx:75
                            eps->side = side_;
                            if (t2_) eps->swap = side_ ? 2 : 1;
                            if (F.resets) {
                                ClothResetRecord *rr_ = F.resets + ((size_t)e * F.n_scripts + n_resets);
                                rr_->init_side = side_;
                                rr_->consumed = 1; rr_->pulls_run = 0; rr_->executed[0] = rr_->executed[1] = rr_->executed[2] = 0;
                                rr_->settle_executed = 0; rr_->tear = 0;
                            }
                        }
                    } else if (tid == 0) {               // episode over and no script left: the slot stays empty
                        ClothStepRecord *r_ = F.records + ((size_t)t_slot * F.E + e);
                        r_->ran = 0; r_->executed = 0; r_->n_grabbed = 0; r_->done = 1; r_->reset_before = 0;
                        eps->t_slot = t_slot + 1;
                    }
                    __syncthreads();
                    continue;
                }
                do_decode = true;
                if (F.policy == CLOTHHIP_POLICY_ORACLE_CORNER) {
                    // examples/analytic.py:105-155 ('distance' method, delta actions): pull the inset corner that is
                    // farthest from its plane corner; candidates in the order ur, lr, ll, ul, the first maximum wins
                    const bool sw = eps->swap == 1;                                       // tier 2, init_side False (:108-114)
                    double best = -1.0;
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int ci = c == 0 ? (sw ? 48 : 598) : (c == 1 ? (sw ? 26 : 576) : (c == 2 ? (sw ? 576 : 26) : (sw ? 598 : 48)));
                        const double tgx = c < 2 ? 1.0 : 0.0, tgy = (c == 0 || c == 3) ? 1.0 : 0.0;
                        const Pt<T> pc = cur[ci];
                        const double x = (double)pc.x, y = (double)pc.y;
                        const double cx = (x - 0.5) * 2.0, cy = (y - 0.5) * 2.0;                 // analytic.py:53-54
                        double dx = tgx - x, dy = tgy - y;                                        // :55-56
                        const double dist = sqrt((x - tgx) * (x - tgx) + (y - tgy) * (y - tgy)); // :57
                        dx = dx * 0.90; dy = dy * 0.90;                                           // :64-66
                        if (dist > best) {
                            best = dist;
                            act[0] = F.ep.clip_act_space ? cx : x; act[1] = F.ep.clip_act_space ? cy : y;   // :151-154
                            act[2] = dx; act[3] = dy;
                        }
                    }
                } else if (with_tier2 && F.policy == CLOTHHIP_POLICY_HIGHEST_POINT) {
                    // examples/analytic.py:792-808: sorted(pts, key=z, reverse=True)[k] -- a stable sort, so equal heights keep
                    // their index order -- with k (the reference: np.random.randint(top_k)) from the caller's table, pulled to
                    // where that point sits on the flat cloth (:742-789). k + 1 rounds of a workgroup arg-max over (z, -index),
                    // each excluding what the earlier rounds took; the per-wave results go through the member list (scratch
                    // between substeps).
                    struct Cand { T z; int i; int pad; };
                    Cand *red = reinterpret_cast<Cand *>(memb);
                    int kc = F.policy_arg[(size_t)(1 + t_slot) * F.E + e];
                    kc = kc < 0 ? 0 : (kc > P - 1 ? P - 1 : kc);
                    T lastz = (T)0; int lasti = -1;
                    const auto better = [](T z1, int i1, T z0, int i0) { return i1 != 0x7fffffff && (i0 == 0x7fffffff || z1 > z0 || (z1 == z0 && i1 < i0)); };
                    for (int round = 0; round <= kc; round++) {
                        T bz = (T)0; int bi = 0x7fffffff;
#pragma unroll
                        for (int q = 0; q < PPT; q++) {
                            const int i = tid + q * NT;
                            if (i < P) {
                                const T z = cur[i].z;
                                const bool ok = lasti < 0 || z < lastz || (z == lastz && i > lasti);
                                if (ok && better(z, i, bz, bi)) { bz = z; bi = i; }
                            }
                        }
                        for (int o = 32; o > 0; o >>= 1) {
                            const T oz = __shfl_xor(bz, o); const int oi = __shfl_xor(bi, o);
                            if (better(oz, oi, bz, bi)) { bz = oz; bi = oi; }
                        }
                        if (lane == 0) { red[tid >> 6].z = bz; red[tid >> 6].i = bi; }
                        __syncthreads();
                        bz = red[0].z; bi = red[0].i;
                        for (int w = 1; w < NT / 64; w++) { const T oz = red[w].z; const int oi = red[w].i; if (better(oz, oi, bz, bi)) { bz = oz; bi = oi; } }
                        lastz = bz; lasti = bi;
                        __syncthreads();
                    }
                    const int pr = lasti / KA_N(&A), pc_ = lasti - pr * KA_N(&A);
                    const Pt<T> pp = cur[lasti];
                    const double x = (double)pp.x, y = (double)pp.y;
                    double tgx, tgy;
                    if (eps->swap == 0) { tgx = F.grid_dx * pr; tgy = F.grid_dy * pc_; }                   // pt.orig_x, pt.orig_y of the flat grid (cloth.pyx:122-124)
                    else { tgx = eps->swap == 2 ? F.grid_dy * pr : 1.0 - F.grid_dy * pr; tgy = F.grid_dx * pc_; }   // :781-788 (orig_z, orig_y)
                    const double cx = (x - 0.5) * 2.0, cy = (y - 0.5) * 2.0;                     // analytic.py:53-54
                    const double dx = (tgx - x) * 0.90, dy = (tgy - y) * 0.90;                    // :55-56, :64-66
                    act[0] = F.ep.clip_act_space ? cx : x; act[1] = F.ep.clip_act_space ? cy : y; // :803-806
                    act[2] = dx; act[3] = dy;
                } else {
                    const double *ap = F.actions + ((size_t)t_slot * F.E + e) * 4;
                    act[0] = ap[0]; act[1] = ap[1]; act[2] = ap[2]; act[3] = ap[3];
                }
            } else {
                if (eps->stop && F.resume != nullptr) {  // the slice ends between two operations of a reset
                    if (tid == 0) {
                        EpResume *rs_ = F.resume + e;
                        rs_->valid = 1; rs_->it = -1; rs_->done_partial = 0; rs_->eps = *eps;
                        if (F.resets != nullptr) {
                            ClothResetRecord *rr_ = F.resets + ((size_t)e * F.n_scripts + n_resets);
                            rs_->rr = *rr_; rr_->consumed = 2;
                        }
                    }
                    break;
                }
                for (;;) {                               // skip the stages this script does not have
                    if (with_tier2 && rp == 8) break;    // tier 2: 1500 updates before the pulls (cloth_env.py:902-903)
                    if (rp < 6) {
                        const int p_ = rp >> 1;
                        if (p_ >= s_n_pulls()) { rp = 6; continue; }
                        if (!(rp & 1) && !s_need_cov(p_)) { rp++; continue; }
                    }
                    if (rp == 6 && s_settle() <= 0) { rp = 7; continue; }
                    break;
                }
                if (with_tier2 && rp == 8) {
                    op = OP_RESET_SETTLE; do_run = true;
                    sc.n_up_end = sc.n_uprest_end = sc.n_pull_end = 0;
                    sc.n_griprest_end = sc.n_total = 1500;
                    sc.break_on_tear = 0;
                } else if (rp < 6 && !(rp & 1)) op = OP_RESET_COND;
                else if (rp < 6) {
                    op = OP_RESET_PULL; do_decode = true;
                    if (rngm) {                          // draw this pull now (cloth_env.py:851-877 tier 1, :959-972 tier 3)
                        if (tid == 0) {
                            ClothResetPull d_;
                            d_.need_coverage = 0; d_.coverage_min = 0.0;
                            if (tier == 1) {
                                d_.point = (int32_t)mt_randint(mt, (uint32_t)P);
                                d_.x = d_.y = 0.0;
                                d_.dx = mt_randval_minabs(mt, -0.20, 0.20, 0.08);
                                d_.dy = mt_randval_minabs(mt, -0.20, 0.20, 0.08);
                                d_.iters_up = F.ep.iters_up;
                            } else if (with_tier2 && tier == 2) {   // cloth_env.py:905-947: hard-coded corner points, no _prevent_oob
                                const double sd = eps->side ? 1.0 : -1.0;
                                d_.need_coverage = 2;     // bit 1: no _prevent_oob
                                d_.x = d_.y = 0.0;
                                d_.iters_up = F.ep.iters_up;
                                if ((rp >> 1) == 0) {
                                    const int ch = mt_double(mt) < 0.5 ? -25 : -1;               // :907
                                    eps->choice = ch;
                                    d_.point = P + ch;
                                    d_.dx = mt_uniform(mt, 0.30, 0.50) * sd;
                                    d_.dy = ch == -25 ? mt_uniform(mt, 0.30, 0.60) : mt_uniform(mt, -0.60, -0.30);
                                } else {
                                    const bool c25 = eps->choice == -25;
                                    d_.point = P + (c25 ? -19 : -7);
                                    d_.dx = mt_uniform(mt, 0.30, 0.60) * sd;
                                    d_.dy = c25 ? mt_uniform(mt, -0.30, -0.60) : mt_uniform(mt, 0.30, 0.60);
                                }
                            } else {
                                d_.iters_up = mt_uniform(mt, 200.0, 280.0);
                                d_.point = -1;
                                d_.x = mt_randval_minabs(mt, 0.30, 0.70, 0.0);
                                d_.y = mt_randval_minabs(mt, 0.30, 0.70, 0.0);
                                d_.dx = mt_randval_minabs(mt, -0.25, 0.25, 0.10);
                                d_.dy = mt_randval_minabs(mt, -0.25, 0.25, 0.10);
                            }
                            eps->pull = d_;
                        }
                        __syncthreads();
                    }
                    const ClothResetPull *pl = rngm ? &eps->pull : &scr->pull[rp >> 1];
                    double px_ = pl->x, py_ = pl->y;
                    const int pt_ = pl->point;
                    if (pt_ >= 0) { const Pt<T> pp = cur[pt_ < P ? pt_ : 0]; px_ = (double)pp.x; py_ = (double)pp.y; }
                    // _prevent_oob (cloth_env.py:834-840)
                    double dx0 = pl->dx, dy0 = pl->dy;
                    if (!(pl->need_coverage & 2)) {
                        if (px_ + dx0 < 0.0) dx0 = 0.0 - px_; else if (px_ + dx0 > 1.0) dx0 = 1.0 - px_;
                        if (py_ + dy0 < 0.0) dy0 = 0.0 - py_; else if (py_ + dy0 > 1.0) dy0 = 1.0 - py_;
                    }
                    // _convert_action_to_clip_space (cloth_env.py:1207-1215), delta actions
                    act[0] = F.ep.clip_act_space ? (px_ - 0.5) * 2 : px_;
                    act[1] = F.ep.clip_act_space ? (py_ - 0.5) * 2 : py_;
                    act[2] = dx0; act[3] = dy0;
                    run_iters_up = pl->iters_up;
                } else if (rp == 6) {
                    op = OP_RESET_SETTLE; do_run = true;
                    sc.n_up_end = sc.n_uprest_end = sc.n_pull_end = 0;
                    sc.n_griprest_end = sc.n_total = s_settle();
                    sc.break_on_tear = 0;
                } else {
                    op = OP_RESET_END;
                }
            }
            int n_grab = 0, iters_pull = 0, decode_err = 0;
            if (do_decode) {
                // ---- action -> schedule (cloth_env.py:396-475), in double, every thread the same arithmetic
                const ClothEpisodeParams &ep = F.ep;
                double a0 = fmax(fmin(act[0], ep.act_high[0]), ep.act_low[0]);                    // :402-415
                double a1 = fmax(fmin(act[1], ep.act_high[1]), ep.act_low[1]);
                const double c2 = fmax(fmin(act[2], ep.act_high[2]), ep.act_low[2]);
                const double c3 = fmax(fmin(act[3], ep.act_high[3]), ep.act_low[3]);
                if (ep.clip_act_space) { a0 = (a0 / 2.0) + 0.5; a1 = (a1 / 2.0) + 0.5; }          // :417-426
                const double tl = sqrt(c2 * c2 + c3 * c3);                                        // :449
                const double xd = c2 / (tl + 1e-5), yd = c3 / (tl + 1e-5);                        // :450-451
                const double xr = xd * ep.reduce_factor, yr = yd * ep.reduce_factor;              // :455-456
                const double stp = sqrt(xr * xr + yr * yr);
                double cl = 0.0;
                int ii = 0;
                for (;;) {                                                                        // :461-468
                    cl = cl + stp;
                    if (cl >= tl) break;
                    ii++;
                    if (ii >= 200000) { decode_err = 1; break; }      // non-finite action: the host wrapper raises
                }
                iters_pull = ii;
                const double iu = run_iters_up;                                                   // :472-475, left to right
                const double b1 = iu, b2 = iu + ep.iters_up_rest, b3 = iu + ep.iters_up_rest + ii;
                const double b4 = iu + ep.iters_up_rest + ii + ep.iters_grip_rest;
                const double b5 = iu + ep.iters_up_rest + ii + ep.iters_grip_rest + ep.iters_rest;
                sc.n_up_end = (int)ceil(b1); sc.n_uprest_end = (int)ceil(b2); sc.n_pull_end = (int)ceil(b3);
                sc.n_griprest_end = (int)ceil(b4); sc.n_total = (int)ceil(b5);
                sc.break_on_tear = 1;
                sc.dz_up = ep.dz_up; sc.dx_pull = xr; sc.dy_pull = yr; sc.dz_pull = 0.0;
                // ---- Gripper.grab_top (gripper.pyx:23-42) on the LDS-resident state, + force_grab (cloth_env.py:434-444)
                const T gx = (T)a0, gy = (T)a1, tt = (T)F.two_thickness;
                double radius = ep.grip_radius;
                for (int tries = 0;; tries++) {
                    const T rad = (T)radius;
                    __syncthreads();
                    if (tid == 0) { misc[8] = 0x7fffffff; misc[9] = 0; }
                    __syncthreads();
                    int best = 0x7fffffff;
                    bool incyl[PPT];
#pragma unroll
                    for (int q = 0; q < PPT; q++) {
                        const int i = tid + q * NT;
                        incyl[q] = false;
                        if (i < P) {
                            const Pt<T> c = cur[i];
                            const T dx = c.x - gx, dy = c.y - gy;
                            if (dx * dx + dy * dy < rad) {                                        // gripper.pyx:35 (radius not squared)
                                incyl[q] = true;
                                for (int l = 0; l < F.n_glevels && l < best; l++) {
                                    T d = c.z - (T)F.levels[l]; d = d < 0 ? -d : d;
                                    if (d < tt) { best = l; break; }                              // gripper.pyx:36
                                }
                            }
                        }
                    }
                    for (int o = 32; o > 0; o >>= 1) { const int v = __shfl_xor(best, o); best = v < best ? v : best; }
                    if (lane == 0 && best != 0x7fffffff) atomicMin(&misc[8], best);
                    __syncthreads();
                    best = misc[8];
                    int n = 0;
                    if (best != 0x7fffffff) {
                        const T lz = (T)F.levels[best];
#pragma unroll
                        for (int q = 0; q < PPT; q++) {
                            if (incyl[q]) {
                                const int i = tid + q * NT;
                                Pt<T> c = cur[i];
                                T d = c.z - lz; d = d < 0 ? -d : d;
                                if (d < tt) {                                                     // pinned = True ; grabbed_pts.append
                                    uint32_t w = w_cnt(c.w);
                                    if ((w & CNT_GRAB_MASK) < CNT_GRAB_MASK) w++;
                                    c.w = w_make<T>(w); cur[i] = c; n++;
                                }
                            }
                        }
                        for (int o = 32; o > 0; o >>= 1) n += __shfl_xor(n, o);
                        if (lane == 0 && n) atomicAdd(&misc[9], n);
                    }
                    __syncthreads();
                    n_grab = misc[9];
                    if (n_grab > 0 || !ep.force_grab || tries >= 10000) break;
                    radius += ep.radius_inc;                                                      // cloth_env.py:439
                }
                do_run = n_grab > 0 && !decode_err;                                               // cloth_env.py:490-493
            }
            // park the plan in LDS: nothing of it stays in registers across the substep loop
            if (tid == 0) {
                eps->rp = rp; eps->op = op; eps->n_grab = n_grab; eps->iters_pull = iters_pull; eps->decode_err = decode_err;
                eps->act[0] = act[0]; eps->act[1] = act[1]; eps->act[2] = act[2]; eps->act[3] = act[3];
            }
            // uniform copies of the schedule for the loop's phase tests
            sc.n_up_end = __builtin_amdgcn_readfirstlane(sc.n_up_end);
            sc.n_uprest_end = __builtin_amdgcn_readfirstlane(sc.n_uprest_end);
            sc.n_pull_end = __builtin_amdgcn_readfirstlane(sc.n_pull_end);
            sc.n_griprest_end = __builtin_amdgcn_readfirstlane(sc.n_griprest_end);
            sc.n_total = __builtin_amdgcn_readfirstlane(do_run ? sc.n_total : 0);
            sc.break_on_tear = __builtin_amdgcn_readfirstlane(sc.break_on_tear);

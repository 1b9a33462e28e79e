// substep_write.inc.hpp -- behind the barrier that ends substep it - 1 (its strain sweep) and the Hooke stage of substep `it`: ClothEnv._pull's
// adjust / release (cloth_env.py:352-367, gripper.pyx:55-73) and the Verlet commit (cloth.pyx:244-256) of substep `it`, by the owner threads.
// A FRAGMENT of k_run_schedule (episode_loop.hpp), included at its place in the kernel body: not a function (see substep_collision.inc.hpp for why).
// Names it uses from the kernel body: pm, Ak_, tid, it, sc, dz_up / dxp / dyp / dzp, cur, misc, pvx/pvy/pvz, nx/ny/nz (substep_hooke_stage.inc.hpp); defines `mode`.
// The owner re-reads its particles' records here (instead of carrying them across the barrier): the record gives the position that becomes the
// previous position (:256), and the pin state AFTER this substep's release.
        int mode = 0; T ax = 0, ay = 0, az = 0;
        if (it < sc.n_up_end) { mode = 1; az = dz_up; }
        else if (it < sc.n_uprest_end) { }
        else if (it < sc.n_pull_end) { mode = 1; ax = dxp; ay = dyp; az = dzp; }
        else if (it < sc.n_griprest_end) { }
        else mode = 2;
        {
            const int P = Ak_->P;
            const bool release_now = mode == 2 && it == sc.n_griprest_end;      // release() is idempotent: only its first call acts
            const bool verlet = (pm & PH_HOOKE) != 0;
            Pt<T> cq[PPT];
#pragma unroll
            for (int q = 0; q < PPT; q++) cq[q] = cur[tid + q * NT < P ? tid + q * NT : 0];     // batched: one LDS latency, not PPT
            if (PIPE && tid == 0) misc[MISC_FRONT] = 0;     // nothing of THIS substep's sweep is finished (the sweeping wave raises it; read again only behind two more barriers)
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                const int i = tid + q * NT;
                if (i >= P) continue;
                Pt<T> c = cq[q];
                uint32_t w = w_cnt(c.w);
                if (release_now && (w & CNT_GRAB_MASK)) {                  // gripper.pyx:68-73: unpinned from this substep on
                    w = 0u;
                    if (!verlet) cur[i].w = w_make<T>(0u);
                }
                const int m = (int)(w & CNT_GRAB_MASK);
                if (mode == 1 && m) {
                    for (int r = 0; r < m; r++) {       // gripper.pyx:60-66: p <- x ; x <- delta + x
                        pvx[q] = c.x; pvy[q] = c.y; pvz[q] = c.z;
                        c.x = ax + c.x; c.y = ay + c.y; c.z = az + c.z;
                    }
                    cur[i] = c;
                } else if (w == 0u && verlet) {         // pinned particles: Verlet skips them (cloth.pyx:244)
                    pvx[q] = c.x; pvy[q] = c.y; pvz[q] = c.z;                                     // :256
                    cur[i] = Pt<T>{nx[q], ny[q], nz[q], w_make<T>(0u)};                           // :255
                }
            }
        }

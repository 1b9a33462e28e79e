// substep_pull.inc.hpp -- ClothEnv._pull between two substeps: Gripper.adjust in the lift / pull phases, Gripper.release once (cloth_env.py:352-367, gripper.pyx:55-73)
// A FRAGMENT of k_run_schedule (episode_loop.hpp), included at its place in the kernel body: not a function. Turning the substep's phases into
// __forceinline__ functions over a context struct was tried (round 5): same instructions, but the register allocation of the 128-VGPR variants
// shifts -- three more scratch reloads in the substep loop, -1.4 % on the headline -- so the split is textual and the ISA is bit-identical to the
// one-file kernel's. Names it uses from the kernel body: it, sc, dz_up, dxp, dyp, dzp, tid, P, cur, pvx/pvy/pvz; defines mode.
        // ---- ClothEnv._pull (cloth_env.py:352-367): adjust / nothing / release -------------------
        int mode = 0; T ax = 0, ay = 0, az = 0;
        if (it < n_up_end_) { mode = 1; az = dz_up; }
        else if (it < n_uprest_end_) { }
        else if (it < n_pull_end_) { mode = 1; ax = dxp; ay = dyp; az = dzp; }
        else if (it < n_griprest_end_) { }
        else mode = 2;
        if (mode == 1) {
            Pt<T> cq[PPT];
#pragma unroll
            for (int q = 0; q < PPT; q++) cq[q] = cur[tid + q * NT < P ? tid + q * NT : 0];     // batched: one LDS latency, not PPT
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                const int i = tid + q * NT;
                if (i >= P) continue;
                Pt<T> c = cq[q];
                const int m = (int)(w_cnt(c.w) & CNT_GRAB_MASK);
                if (m) {
                    for (int r = 0; r < m; r++) {       // gripper.pyx:60-66: p <- x ; x <- delta + x
                        pvx[q] = c.x; pvy[q] = c.y; pvz[q] = c.z;
                        c.x = ax + c.x; c.y = ay + c.y; c.z = az + c.z;
                    }
                    cur[i] = c;
                }
            }
            __syncthreads();
        } else if (mode == 2 && it == n_griprest_end_) {      // release() is idempotent: only its first call acts
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                const int i = tid + q * NT;
                if (i >= P) continue;
                const uint32_t c = w_cnt(cur[i].w);
                if (c & CNT_GRAB_MASK) cur[i].w = w_make<T>(0u);    // gripper.pyx:68-73
            }
            __syncthreads();
        }


// substep_plane.inc.hpp -- plane collision (cloth.pyx:345-370), by the owner (it holds the previous position)
// A FRAGMENT of k_run_schedule (episode_loop.hpp), included at its place in the kernel body: not a function. Turning the substep's phases into
// __forceinline__ functions over a context struct was tried (round 5): same instructions, but the register allocation of the 128-VGPR variants
// shifts -- three more scratch reloads in the substep loop, -1.4 % on the headline -- so the split is textual and the ISA is bit-identical to the
// one-file kernel's. Names it uses from the kernel body: pm, Ak_, tid, cur, pvx/pvy/pvz, misc.
        // ---- plane (cloth.pyx:345-370), by the owner (it holds the previous position) --------------------
        if (pm & PH_PLANE) {
            CLOTH_PHASE_ARGS()
            const T k_min_z = k.min_z, k_surf_off = k.surf_off, k_one_m_fric = k.one_m_fric;
            Pt<T> mq[PPT];
#pragma unroll
            for (int q = 0; q < PPT; q++) mq[q] = cur[tid + q * NT < P ? tid + q * NT : 0];     // batched: one LDS latency, not PPT
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                const int i = tid + q * NT;
                if (i >= P) continue;
                const Pt<T> me = mq[q];
#ifdef CLOTHHIP_CELL_COUNTERS
                if (!w_cnt(me.w) && me.z >= k_min_z) atomicOr(&misc[12], 1);     // census: an unpinned particle the plane did not restore
#endif
                if (w_cnt(me.w) || me.z >= k_min_z) continue;
                const T px = pvx[q], py = pvy[q], pz = pvz[q];
                const T t = (k_min_z - pz) * (T)1.0;
                const T tgx = px + t * (T)(-0.0), tgy = py + t * (T)(-0.0), tgz = pz + t * (T)(-1.0);
                const T gx = tgx + k_surf_off * (T)0.0, gy = tgy + k_surf_off * (T)0.0, gz = tgz + k_surf_off * (T)1.0;
                const T cx = gx - px, cy = gy - py, cz = gz - pz;
                cur[i] = Pt<T>{mad<T>(cx, k_one_m_fric, px), mad<T>(cy, k_one_m_fric, py), mad<T>(cz, k_one_m_fric, pz), me.w};
            }
        }

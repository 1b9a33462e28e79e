// substep_hooke_verlet.inc.hpp -- gravity + Hooke + Verlet (cloth.pyx:216-256): per-particle gather in ascending list index
// A FRAGMENT of k_run_schedule (episode_loop.hpp), included at its place in the kernel body: not a function. Turning the substep's phases into
// __forceinline__ functions over a context struct was tried (round 5): same instructions, but the register allocation of the 128-VGPR variants
// shifts -- three more scratch reloads in the substep loop, -1.4 % on the headline -- so the split is textual and the ISA is bit-identical to the
// one-file kernel's. Names it uses from the kernel body: pm, Ak_, tid, cur, pvx/pvy/pvz, gt, rr, vm, lean_entry, lean_rest, rest_at.
        // ---- gravity + Hooke gather + Verlet (cloth.pyx:216-256) ----------------------------------
        if (pm & PH_HOOKE) {
            CLOTH_PHASE_ARGS()
            // Per particle: f = (0,0,m*g) + sum over its incident springs in ascending list index of fm * (nbr - self).
            // (For the spring's ptB the reference adds -(fm * (self - nbr)), which is the same IEEE value.)
            // Branch-free: absent slots (grid border) and pinned particles are computed and discarded.
            T nx[PPT], ny[PPT], nz[PPT];
            uint32_t wme[PPT];
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                nx[q] = ny[q] = nz[q] = (T)0; wme[q] = 1u;
                // a real branch per particle: each particle's 12 springs form their own scheduling region, which
                // keeps the register allocator from interleaving all PPT*12 spring evaluations at once
                if (tid + q * NT < P) {
                    const Pt<T> me = cur[tid + q * NT];
                    wme[q] = w_cnt(me.w);
                    T fx = (T)0 + (T)0, fy = (T)0 + (T)0, fz = (T)0 + k.mg;
                    uint32_t gl[HK_SLOTS];
                    int iq_ = tid + q * NT; uint4 lw_ = uint4{0u, 0u, 0u, 0u}; if constexpr (LEAN64) lw_ = lstc[iq_]; uint32_t vq_ = LEAN64 ? lw_.x : vm[(LEAN && !LEAN64) ? q : 0];
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
                        const T r = LEAN64 ? lean_rest64(sl, lw_) : LEAN ? lean_rest(sl) : (REST_R ? rr[REST_R ? q : 0][sl] : rest_at((g >> HK_POS_SHIFT) & HK_POS_MASK));
                        const T kk = (LEAN ? lean_bend(sl) : (g & HK_BEND) != 0u) ? k.ks_bend : k.ks_str;
                        const T dx = nb.x - me.x, dy = nb.y - me.y, dz = nb.z - me.z;
                        const T l = fastnorm<T>(dx, dy, dz);                                      // :231
                        const T fm = dev_div<T>(kk * (l - r), l);                                 // :232
                        const bool valid = (g & HK_VALID) != 0u;
                        fx = valid ? mad<T>(fm, dx, fx) : fx; fy = valid ? mad<T>(fm, dy, fy) : fy; fz = valid ? mad<T>(fm, dz, fz) : fz;   // :236-237
                    }
                    nx[q] = mad<T>(fx, k.dsm, mad<T>(k.damp, me.x - pvx[q], me.x));               // :249
                    ny[q] = mad<T>(fy, k.dsm, mad<T>(k.damp, me.y - pvy[q], me.y));
                    nz[q] = mad<T>(fz, k.dsm, mad<T>(k.damp, me.z - pvz[q], me.z));
                    if (wme[q] == 0) { pvx[q] = me.x; pvy[q] = me.y; pvz[q] = me.z; }             // :256
                }
            }
            __syncthreads();                                // every neighbour read of the old positions is done
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                const int i = tid + q * NT;
                if (wme[q]) continue;                       // pinned (or no particle): Verlet skips it (cloth.pyx:244)
                cur[i] = Pt<T>{nx[q], ny[q], nz[q], w_make<T>(0u)};                               // :255
            }
        }


// substep_collision.inc.hpp -- spatial map + self-collision (cloth.pyx:298-343): LDS hash build, seed test, exact Gauss-Seidel cell sweeps by LDS tickets (the cell kernels of phase_collide.hpp); RELAXED builds: Jacobi order
// A FRAGMENT of k_run_schedule (episode_loop.hpp), included at its place in the kernel body: not a function. Turning the substep's phases into
// __forceinline__ functions over a context struct was tried (round 5): same instructions, but the register allocation of the 128-VGPR variants
// shifts -- three more scratch reloads in the substep loop, -1.4 % on the headline -- so the split is textual and the ISA is bit-identical to the
// one-file kernel's. Names it uses from the kernel body: pm, Ak_, tid, lane, cur, hkey, hco, memb, slot, misc, olist, alist_end, cpos, smem, lay, tph / TSTAMP.
        // ---- spatial map (cloth.pyx:298-311): hash table in LDS keyed by the exact cell key + a list of the occupied
        // slots; members of a cell are stored contiguously (CSR); ascending point index is restored by the sweep.
        if (pm & PH_COLLIDE) {
            CLOTH_PHASE_ARGS()
            uint32_t ch[PPT], rank[PPT];
            {
                uint32_t ckey[PPT];
                bool pend[PPT], made[PPT];
                bool anyp = false;
#pragma unroll
                for (int q = 0; q < PPT; q++) {             // batched: the PPT particles' LDS traffic overlaps
                    const int i = tid + q * NT;
                    const Pt<T> c = cur[i < P ? i : 0];                                           // own slot: no hazard
                    ckey[q] = cell_key<T>(k, c.x, c.y, c.z);
#ifdef CLOTHHIP_CELL_COUNTERS
                    if (i < P) {      // census: did any particle change its cell since the previous substep?
                        uint32_t *lk_ = reinterpret_cast<uint32_t *>(smem + lay.lkey);
                        if (lk_[i] != ckey[q]) atomicOr(&misc[16], 1);
                        lk_[i] = ckey[q];
                    }
#endif
                    // (ht_bits 0: a table whose size is not a power of two -- the two-per-CU layout of the large grids -- is indexed by the
                    //  high half of hash x size; which slot a cell gets never shows in the results)
                    ch[q] = KA_HTBITS(Ak_) ? (ckey[q] * 2654435761u) >> (32 - KA_HTBITS(Ak_)) : __umulhi(ckey[q] * 2654435761u, (uint32_t)HT);
                    pend[q] = i < P; made[q] = false; anyp |= pend[q];
                }
                // linear probing; the table has >= 1.5 P slots, so a free one always exists -- the probe bound only
                // guarantees termination should LDS ever be corrupted
                for (int probe = 0; anyp && probe < HT; probe++) {
                    anyp = false;
#pragma unroll
                    for (int q = 0; q < PPT; q++) {
                        if (pend[q]) {
                            const uint32_t old = atomicCAS(&hkey[ch[q]], KEY_EMPTY, ckey[q]);
                            if (old == KEY_EMPTY || old == ckey[q]) { pend[q] = false; made[q] = old == KEY_EMPTY; }
                            else { ch[q] = ch[q] + 1u >= (uint32_t)HT ? 0u : ch[q] + 1u; anyp = true; }
                        }
                    }
                }
                int nmade = 0;
#pragma unroll
                for (int q = 0; q < PPT; q++) {
                    const int i = tid + q * NT;
                    rank[q] = 0;
                    if (i < P) { slot[i] = (uint16_t)ch[q]; rank[q] = atomicAdd(&hco[ch[q]], 1u); }   // my place in the cell
                    nmade += made[q] ? 1 : 0;
                }
                // whoever created a slot lists it: one LDS atomic per wave
                const int inc = wave_incl_scan(nmade);
                const int tot = __builtin_amdgcn_readlane(inc, 63);
                if (tot) {
                    int base = 0;
                    if (lane == 0) base = atomicAdd(&misc[3], tot);
                    int o = __builtin_amdgcn_readfirstlane(base) + inc - nmade;
#pragma unroll
                    for (int q = 0; q < PPT; q++)
                        if (made[q]) olist[o++] = (uint16_t)ch[q];
                }
            }
            __syncthreads();
            TSTAMP(2)
            const int nocc = __builtin_amdgcn_readfirstlane(misc[3]);
            for (int t0 = 0; t0 < nocc; t0 += NT) {         // member range of every occupied cell (any order)
                const int t = t0 + tid;
                const int h = t < nocc ? (int)olist[t] : 0;
                const int c = t < nocc ? (int)hco[h] : 0;
                const int inc = wave_incl_scan(c);
                int base = 0;
                if (lane == 63) base = atomicAdd(&misc[4], inc);
                base = __builtin_amdgcn_readlane(base, 63);
                if (t < nocc) hco[h] = ((uint32_t)(base + inc - c) << 16) | (uint32_t)c;          // (start << 16) | count
            }
            __syncthreads();
            TSTAMP(3)
            int cn[PPT], cstart[PPT];
            Pt<T> cme[PPT];
            int nmax = 0;
#pragma unroll
            for (int q = 0; q < PPT; q++) {
                const int i = tid + q * NT;
                const uint32_t co = hco[ch[q]];
                cme[q] = cur[i < P ? i : 0];
                cstart[q] = (int)(co >> 16);
                if (i < P) {
                    memb[cstart[q] + (int)rank[q]] = (uint16_t)i;
                    if (KA_CELLCOPY(Ak_)) cpos[cstart[q] + (int)rank[q]] = Pt<T>{cme[q].x, cme[q].y, cme[q].z, w_make<T>((uint32_t)i)};
                }
                const bool use = i < P && w_cnt(cme[q].w) == 0;
                cn[q] = use ? (int)(co & 0xFFFFu) : 0;
                if (cn[q] < 2) cn[q] = 0;
                nmax = cn[q] > nmax ? cn[q] : nmax;
            }
            __syncthreads();
            TSTAMP(4)
            if constexpr (RELAXED) {
                // Jacobi order (NOT the reference's Gauss-Seidel order, cloth.pyx:313-343): every unpinned particle collects its hits
                // against the cell-ordered snapshot of the phase's start positions and moves by itself; no seeds, no cell sweeps
#pragma unroll
                for (int q = 0; q < PPT; q++) {
                    const int i = tid + q * NT;
                    if (cn[q] >= 2) {
                        const Pt<T> me_ = cme[q];
                        T tx = (T)0, ty = (T)0, tz = (T)0; int nh = 0;
#pragma unroll 1
                        for (int b = 0; b < cn[q]; b++) {
                            const Pt<T> o = cpos[cstart[q] + b];
                            const T dx = me_.x - o.x, dy = me_.y - o.y, dz = me_.z - o.z;
                            const T dist = dev_sqrt<T>(sumsq<T>(dx, dy, dz));
                            const bool hit_ = ((int)w_cnt(o.w) != i) & (dist <= k.thresh);
                            const T factor = hit_ ? dev_div<T>(k.thresh - dist, dist) : (T)0;
                            tx = hit_ ? mad<T>(dx, factor, tx) : tx; ty = hit_ ? mad<T>(dy, factor, ty) : ty; tz = hit_ ? mad<T>(dz, factor, tz) : tz;
                            nh += hit_ ? 1 : 0;
                        }
                        if (nh) {
                            const T nf = (T)nh;
                            cur[i] = Pt<T>{me_.x + dev_div<T>(dev_div<T>(tx, nf), k.sim_steps), me_.y + dev_div<T>(dev_div<T>(ty, nf), k.sim_steps),
                                           me_.z + dev_div<T>(dev_div<T>(tz, nf), k.sim_steps), me_.w};
                        }
                    }
                }
            } else {
            // ---- self-collision (cloth.pyx:313-343) ------------------------------------------------------
            // (1) seeds: every unpinned particle, in parallel: does it have a hit (a same-cell member within
            //     2*thickness) at the CURRENT positions? A seed gets the flag bit of its slot word; the first seed of
            //     a cell (whoever wins) puts the cell on the active list. Conservative by the filter slack; the sweep
            //     re-tests exactly. With the cell-ordered record copy a pair costs ONE LDS read.
            {
                const T thr2 = k.thresh * k.thresh * ((T)1 + filt_slack<T>());
                bool hit[PPT];
#pragma unroll
                for (int q = 0; q < PPT; q++) hit[q] = false;
                // (left to itself the compiler unrolls the member loops several times: fine at 256 VGPRs, 500 spilled registers at
                //  the LEAN variant's 168 -- that variant gets its own copy of the loops, not unrolled)
#ifndef CLOTHHIP_BISECT_NOPRECHECK
                if constexpr (LEAN || (NT == 512 && PPT == 2)) {
                    // register-lean form (the builds with a VGPR cap: LEAN, eight waves per cloth): one owned particle after the other (a real branch each: a scheduling region of its own),
                    // the member loop not unrolled; the trip count is the wave's largest member count for THAT particle
#pragma unroll
                    for (int q = 0; q < PPT; q++) {
                        const int nq = -__builtin_amdgcn_readlane(wave_incl_min(-cn[q]), 63);
                        if (nq > 0) {
                            int iq_ = tid + q * NT;
                            asm volatile("" : "+v"(iq_));
                            const Pt<T> me_ = cur[iq_ < P ? iq_ : 0];
                            const int cs_ = cstart[q], cn_ = cn[q];
                            bool h_ = false;
                            if (KA_CELLCOPY(Ak_)) {
                                // a read past the cell's range (another cell's record or the padding behind the array) is masked out
                                // by the member count; the trip base is clamped so that no read leaves the padded array
                                // (round 5: four members per trip -- half the loop branches and LDS waits per member: +0.25 % at two cloths per
                                //  CU; two per trip where the register cap is 80 (six per CU): 23 fewer spill reloads, +1.6 % at 1536 cloths)
                                if constexpr (TAB > CLOTHHIP_PRECHECK2_MAX_TAB) {
#pragma unroll 1
                                for (int b = 0; b < nq; b += 4) {
                                    const int base = cs_ + b < Ppad + 28 ? cs_ + b : Ppad + 28;
                                    const Pt<T> o0 = cpos[base], o1 = cpos[base + 1], o2 = cpos[base + 2], o3 = cpos[base + 3];
                                    const T dx0 = me_.x - o0.x, dy0 = me_.y - o0.y, dz0 = me_.z - o0.z;
                                    const T dx1 = me_.x - o1.x, dy1 = me_.y - o1.y, dz1 = me_.z - o1.z;
                                    const T dx2 = me_.x - o2.x, dy2 = me_.y - o2.y, dz2 = me_.z - o2.z;
                                    const T dx3 = me_.x - o3.x, dy3 = me_.y - o3.y, dz3 = me_.z - o3.z;
                                    h_ |= (b < cn_) & ((int)w_cnt(o0.w) != iq_) & !(sumsq<T>(dx0, dy0, dz0) > thr2);
                                    h_ |= (b + 1 < cn_) & ((int)w_cnt(o1.w) != iq_) & !(sumsq<T>(dx1, dy1, dz1) > thr2);
                                    h_ |= (b + 2 < cn_) & ((int)w_cnt(o2.w) != iq_) & !(sumsq<T>(dx2, dy2, dz2) > thr2);
                                    h_ |= (b + 3 < cn_) & ((int)w_cnt(o3.w) != iq_) & !(sumsq<T>(dx3, dy3, dz3) > thr2);
                                }
                                } else {
#pragma unroll 1
                                for (int b = 0; b < nq; b += 2) {
                                    const int base = cs_ + b < Ppad + 30 ? cs_ + b : Ppad + 30;
                                    const Pt<T> o0 = cpos[base], o1 = cpos[base + 1];
                                    const T dx0 = me_.x - o0.x, dy0 = me_.y - o0.y, dz0 = me_.z - o0.z;
                                    const T dx1 = me_.x - o1.x, dy1 = me_.y - o1.y, dz1 = me_.z - o1.z;
                                    h_ |= (b < cn_) & ((int)w_cnt(o0.w) != iq_) & !(sumsq<T>(dx0, dy0, dz0) > thr2);       // branch-free on purpose (& not &&)
                                    h_ |= (b + 1 < cn_) & ((int)w_cnt(o1.w) != iq_) & !(sumsq<T>(dx1, dy1, dz1) > thr2);
                                }
                                }
                            } else {
#pragma unroll 1
                                for (int b = 0; b < nq; b += 2) {
                                    const int j0 = (int)memb[cn_ ? cs_ + (b < cn_ ? b : 0) : 0], j1 = (int)memb[cn_ ? cs_ + (b + 1 < cn_ ? b + 1 : 0) : 0];
                                    const Pt<T> o0 = cur[j0], o1 = cur[j1];
                                    const T dx0 = me_.x - o0.x, dy0 = me_.y - o0.y, dz0 = me_.z - o0.z;
                                    const T dx1 = me_.x - o1.x, dy1 = me_.y - o1.y, dz1 = me_.z - o1.z;
                                    h_ |= (b < cn_) & (j0 != iq_) & !(sumsq<T>(dx0, dy0, dz0) > thr2);
                                    h_ |= (b + 1 < cn_) & (j1 != iq_) & !(sumsq<T>(dx1, dy1, dz1) > thr2);
                                }
                            }
                            hit[q] = h_;
                        }
                    }
                } else {
                    if (KA_CELLCOPY(Ak_)) {
                        constexpr int CU = 2;
                        // a read past the cell's range (another cell's record or the padding behind the array) is masked out
                        // by the member count; the trip base is clamped so that no read leaves the padded array
                        for (int b = 0; b < nmax; b += CU) {     // CU members x PPT particles per trip: their LDS reads overlap
                            Pt<T> o[PPT][CU];
#pragma unroll
                            for (int q = 0; q < PPT; q++) {
                                const int base = cstart[q] + b < Ppad + 32 - CU ? cstart[q] + b : Ppad + 32 - CU;
#pragma unroll
                                for (int u = 0; u < CU; u++) o[q][u] = cpos[base + u];
                            }
#pragma unroll
                            for (int q = 0; q < PPT; q++)
#pragma unroll
                                for (int u = 0; u < CU; u++) {                              // branch-free on purpose (& not &&)
                                    const T dx = cme[q].x - o[q][u].x, dy = cme[q].y - o[q][u].y, dz = cme[q].z - o[q][u].z;
                                    const bool other = (b + u < cn[q]) & ((int)w_cnt(o[q][u].w) != tid + q * NT);
                                    hit[q] |= other & !(sumsq<T>(dx, dy, dz) > thr2);
                                }
                        }
                    } else {
                        for (int b = 0; b < nmax; b += 4) {
                            int jj[PPT][4];
#pragma unroll
                            for (int q = 0; q < PPT; q++)
#pragma unroll
                                for (int u = 0; u < 4; u++) {
                                    const int bb = b + u < cn[q] ? b + u : 0;
                                    jj[q][u] = (int)memb[cn[q] ? cstart[q] + bb : 0];
                                }
#pragma unroll
                            for (int q = 0; q < PPT; q++)
#pragma unroll
                                for (int u = 0; u < 4; u++) {
                                    const Pt<T> o = cur[jj[q][u]];
                                    const T dx = cme[q].x - o.x, dy = cme[q].y - o.y, dz = cme[q].z - o.z;
                                    const bool other = (b + u < cn[q]) & (jj[q][u] != tid + q * NT);
                                    hit[q] |= other & !(sumsq<T>(dx, dy, dz) > thr2);
                                }
                        }
                    }
                }
#endif
#pragma unroll
                for (int q = 0; q < PPT; q++) {
                    if (hit[q]) {
                        slot[tid + q * NT] = (uint16_t)(ch[q] | 0x8000u);
                        if (atomicMin(&hkey[ch[q]], (uint32_t)(tid + q * NT)) >= KEY_FLOOR)
                            alist_end[-atomicAdd(&misc[2], 1)] = (uint16_t)ch[q];
                    }
                }
            }
            __syncthreads();
            TSTAMP(5)
            // (2) the active cells (those with a seed): exact Gauss-Seidel sweep (cells are independent: each particle
            // sits in exactly one). Every wave reads the whole list; work is handed out by LDS tickets so that the waves
            // finish together: first the cells with more than 16 members, one per wave at a time, then the small
            // cells four at a time, one per 16-lane group (two larger cells per wave in 32-lane groups was measured:
            // the bpermute broadcasts cost what the pairing saves).
            {
                const int na = __builtin_amdgcn_readfirstlane(misc[2]);
#ifdef CLOTHHIP_CELL_COUNTERS
                tph[9] += 64 * na; tph[10] += 64 * nocc; tph[0] += na == 0 ? 64 : 0;
                tph[7] += 64 * (-__builtin_amdgcn_readlane(wave_incl_min(-nmax), 63));
#endif
                if (na) __builtin_amdgcn_s_setprio(2);     // serial per-cell sweeps: latency-critical like the strain sweep
                int tkb = -1, tks = -1, bbase = 0, sbase = 0;       // outstanding tickets, tickets used up by earlier chunks
                for (int c0 = 0; c0 < na; c0 += 64) {
                    const int ei = c0 + lane;
                    const bool ev = ei < na;
                    const int hs_l = ev ? (int)alist_end[-ei] : 0;
                    const uint32_t co_l = ev ? hco[hs_l] : 0u;
                    const int n_l = (int)(co_l & 0xFFFFu);
                    unsigned long long big = ballot64(ev && n_l > 16);
                    unsigned long long sm = ballot64(ev && n_l <= 16);
                    const int nbig = (int)__popcll(big), nsb = ((int)__popcll(sm) + 3) >> 2;
                    for (int used = 0;;) {
                        if (tkb < 0) { int t = 0; if (lane == 0) t = atomicAdd(&misc[5], 1); tkb = __builtin_amdgcn_readfirstlane(t); }
                        if (tkb >= bbase + nbig) break;             // that ticket is for a later chunk (or nothing)
                        for (; used < tkb - bbase; used++) big &= big - 1ull;
                        tkb = -1;
                        const int b = __builtin_amdgcn_readfirstlane(__ffsll((long long)big) - 1);
                        const uint32_t co = (uint32_t)__builtin_amdgcn_readlane((int)co_l, b);
                        const int n = (int)(co & 0xFFFFu);
                        uint16_t *m = memb + (int)(co >> 16);
                        if (n <= 64) {
#ifndef CLOTHHIP_BISECT_NOWAVE
                            const int nv_ = collide_cell_wave<T>(cur, m, slot, n, k, lane);
#else
                            const int nv_ = 0;
#endif
#ifdef CLOTHHIP_CELL_COUNTERS
                            tph[4] += 64; tph[5] += 64 * n; tph[6] += 64 * (nv_ & 0xffff); tph[11] += 64 * (nv_ >> 16);
#else
                            (void)nv_;
#endif
                        } else if (lane == 0) collide_cell_serial<T>(cur, m, n, k);
                    }
                    bbase += nbig;
#ifdef CLOTHHIP_CELL_STAMPS
                    TSTAMP(10)
#endif
                    for (int used = 0;;) {                        // up to four small cells per ticket
                        if (tks < 0) { int t = 0; if (lane == 0) t = atomicAdd(&misc[6], 1); tks = __builtin_amdgcn_readfirstlane(t); }
                        if (tks >= sbase + nsb) break;
                        for (; used < 4 * (tks - sbase); used++) sm &= sm - 1ull;
                        tks = -1;
                        int hs = -1;
#pragma unroll
                        for (int g = 0; g < 4; g++) {
                            if (sm) {
                                const int b = __builtin_amdgcn_readfirstlane(__ffsll((long long)sm) - 1);
                                sm &= sm - 1ull;
                                const int v = __builtin_amdgcn_readlane(hs_l, b);
                                hs = (lane >> 4) == g ? v : hs;
                            }
                        }
                        used += 4;
#ifdef CLOTHHIP_CELL_COUNTERS
                        tph[8] += 64;
#endif
#ifndef CLOTHHIP_BISECT_NOGROUP
                        collide_cells_group<T, 16>(cur, memb, slot, hco, hs, k, lane);
#endif
                    }
                    sbase += nsb;
#ifdef CLOTHHIP_CELL_STAMPS
                    TSTAMP(11)
#endif
                }
            }
            __builtin_amdgcn_s_setprio(0);
            }   // (exact order)
            __syncthreads();
            TSTAMP(6)
            for (int t = tid; t < nocc; t += NT) { const int h = (int)olist[t]; hkey[h] = KEY_EMPTY; hco[h] = 0; }   // ready for the next substep
            if (tid == 0) { misc[2] = 0; misc[3] = 0; misc[4] = 0; misc[5] = 0; misc[6] = 0; }
        } else {
            __syncthreads();
        }


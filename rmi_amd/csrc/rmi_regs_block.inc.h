// The body of one block of 16 steps of k_leaf_regs' walk (rmi_regs.hip.h includes this text twice: STATIC = true inside a loop
// the compiler unrolls completely -- `b` is then a constant of every copy --, STATIC = false inside the rolled loop).  A text
// and not a lambda: called through one, the very same body compiles to a loop a third longer, with part of the stash in
// scratch memory (the inliner's order decides what the register allocator sees).  Context: b, STATIC, cur, npts, xs, the running
// sums and their copies, rawA / rawB, cA / cB and the lambdas of the kernel.
        unsigned int in_b, dlt;
        block_base(b, in_b, dlt);
        XT T[8];                                                         // (F32: already x - x0 as a float)
        // Masking: a lane is finished behind its container's last point, and stays finished.  Its sums are put aside at the end
        // of the half block (8 steps) in which it finishes; from then on it may compute what it likes.  So a half block in
        // which every lane is either alive for all 8 steps or finished before the first -- the rule while the walk is younger
        // than the shortest container -- runs WITHOUT any test; else every step tests.  (Tried: narrowing EXEC once per step
        // by v_cmpx behind the compiler's back -- cheap, but every copy or spill the register allocator places inside such a
        // region moves only the lanes still alive, and at 500 registers it places them.)
        // The constants of a half block come through the scalar cache, and scalar loads share their counter with the LDS reads
        // without returning in order: a wait for them is a wait for every LDS read in flight.  So they are asked for FIRST and
        // waited for at once (a hit in the scalar cache: tens of cycles), together with this half's keys, asked for a half block
        // ago; only then are the next half's keys requested, and those land under the arithmetic.
        // The constants of the steps come through the scalar cache, a quarter block (4 steps) at a time and one quarter AHEAD, in two
        // alternating sets (cA, cB: 24 SGPRs each).  Scalar loads share their counter with the LDS reads without returning in order,
        // so a wait for them is a wait for every LDS read in flight: the waits stand where the keys asked for a half block ago are
        // needed anyway (a half's start) and in the middle of a half, 4 steps behind the next half's key requests.
        // (the builtin, not an asm statement: the compiler keeps its own score of the LDS reads and scalar loads in flight, and
        //  what it cannot see waited for it waits for again -- with lgkmcnt(0) at every step while a scalar load is out)
        auto landed = [&]() { __builtin_amdgcn_s_waitcnt(0xC07F); asm volatile("" ::: "memory"); };   // vmcnt 63, expcnt 7, lgkmcnt 0
        auto quarter = [&](int hb, int qr, const RAW (&raw)[8], const double (&c)[12], auto full_tag) {
          constexpr bool FULL = decltype(full_tag)::value;
          const unsigned int k0 = b * (unsigned int)RG_ROW + (unsigned int)(hb * 8 + qr * 4);
#pragma unroll
          for (int u = 0; u < 4; u++) {
            const int q = qr * 4 + u;
            const unsigned int k = k0 + (unsigned int)u;
            const double x = rg_as_float<K>(raw[q]);
            if constexpr (F32) {
              if (rg_block_index(b) == 0u && hb == 0 && q == 0) x0 = x;   // (the container's first key)
              T[q] = (float)(x - x0);
            } else if constexpr (W2) T[q] = rg_raw_lo(raw[q]);
            else T[q] = x;
            if (FULL || k < npts) {
              if (!(RG_DIAG & 2)) {
                if constexpr (DIVK) { if (x == xp) dmin = 0u; }
                else { const unsigned int d = rg_raw_lo(raw[q]) ^ plo; dmin = dmin < d ? dmin : d; }
              }
              if (!(RG_KO & 1)) {
                const double dx = x - mx;                                 // linear.rs:26
                if constexpr (DIVK) mx += dx / c[4 + u]; else mx += div_by_count2(dx, c[u], c[4 + u]);   // :27
                cc += dx * c[8 + u];                                      // :28-29 in closed form (head of rmi_lanes.hip.h)
                m2 += dx * (x - mx);                                      // :30-31
              }
            }
            xp = x; plo = rg_raw_lo(raw[q]);
          }
          // (written out, the blocks are separated only by the branches around the panel requests, and the compiler sinks the
          //  whole arithmetic behind the last of them -- with every quarter's 24 constants parked in VGPR lanes meanwhile)
          if constexpr (STATIC) asm volatile("" : "+v"(mx), "+v"(cc), "+v"(m2), "+v"(dmin));
        };
        // a half block's doubles to their registers of the stash
        auto stash_half = [&](int hb) {
          // (a chain of tests in three groups of four: as a `switch` the cases meet in one block of phis, and the register
          //  allocator then shuffles the whole stash around in every case)
          auto group = [&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            rg_static_for<4 * g, (4 * g + 4 < SBLK ? 4 * g + 4 : SBLK)>([&](auto i_tag) {
              constexpr int i = decltype(i_tag)::value;
              if (b == (unsigned int)i && !((RG_DIAG & 16) && i > 0)) {
#pragma unroll
                for (int q = 0; q < 8; q++) xs[i * RG_ROW + hb * 8 + q] = T[q];
                asm volatile("; stash bank %0" ::"n"(i));                 // (keeps the cases apart: merged, xs[] would be indexed by b, i.e. memory)
              }
            });
          };
          rg_static_for<0, (SBLK + 3) / 4>([&](auto g_tag) {
            constexpr int g = decltype(g_tag)::value;
            if (b >= (unsigned int)(4 * g) && b < (unsigned int)(4 * g + 4)) group(g_tag);
          });
        };
        auto run_half = [&](int hb, const RAW (&raw)[8], auto&& prefetch) {
          const unsigned int k0 = b * (unsigned int)RG_ROW + (unsigned int)(hb * 8);
          const bool full = STATIC || (RG_DIAG & 8) || __all(npts >= k0 + 8u || npts <= k0);
          sub0();
          landed();                                                      // (cA and this half's keys)
          request(cB, 4u * b + 2u * (unsigned int)hb + 1u);
          sub(9);
          prefetch();                                                    // (the next half's keys: they land under the arithmetic)
          if (full) quarter(hb, 0, raw, cA, std::true_type{}); else quarter(hb, 0, raw, cA, std::false_type{});
          sub(8);
          // (the walk's last steps: where the longest container ends in a half's first quarter the second is not run -- with
          //  evenly filled leaves EVERY group's last block holds one or two steps)
          if (STATIC || k0 + 4u < cur.maxlen) {
            landed();                                                    // (cB)
            request(cA, 4u * b + 2u * (unsigned int)hb + 2u);
            sub(9);
            if (full) quarter(hb, 1, raw, cB, std::true_type{}); else quarter(hb, 1, raw, cB, std::false_type{});
            sub(8);
          } else {
#pragma unroll
            for (int q = 4; q < 8; q++) T[q] = (XT)0;                    // (stashed with the others; no leaf has these steps)
          }
          if constexpr (!STATIC) {
            const bool ends = npts > k0 && npts <= k0 + 8u;
            if (__any(ends)) { if (ends) { fmx = mx; fcc = cc; fm2 = m2; fdmin = dmin; } }
          }
          stash_half(hb);
          sub(7);
        };
        // first half: its keys were asked for a half block ago; the second half's are asked for now
        run_half(0, rawA, [&]() {
#pragma unroll
          for (int q = 0; q < 8; q++) rawB[q] = slot_key(in_b, dlt, 8 + q);
        });
        // second half.  Every read of panel b is behind us: its ring slot takes panel b + 4 -- but nothing behind the walk's
        // last panel, so that at the end of the fit the ring still holds the tail of every row (the steps >= RG_STASH of the
        // error pass).  Then the first keys of the next block: panel b + 2 has landed once at most the panels behind it are
        // outstanding.
        if (STATIC || b * (unsigned int)RG_ROW + 8u < cur.maxlen)
        run_half(1, rawB, [&]() {
          if (!(RG_KO & 4) && b + (unsigned int)RG_RING <= cur.lastp) issue_panel(cur.kb, b + (unsigned int)RG_RING, cur.off, cur.lim);
          sub(5);
          if ((b + 1u) * (unsigned int)RG_ROW < cur.maxlen) {
            if (!(RG_KO & 4)) {
              if (cur.lastp >= b + 4u) rg_wait_vm<2 * NI>();
              else if (cur.lastp == b + 3u) rg_wait_vm<NI>();
              else rg_wait_vm<0>();
            }
            sub(3 + 8);
            unsigned int nx_b, nx_d;
            block_base(b + 1u, nx_b, nx_d);
#pragma unroll
            for (int q = 0; q < 8; q++) rawA[q] = slot_key(nx_b, nx_d, q);
          }
        });

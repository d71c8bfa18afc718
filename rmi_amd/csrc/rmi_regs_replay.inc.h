// The replay of the stash's banks [b_from, b_to) in k_leaf_regs' error pass (rmi_regs.hip.h includes this text once per instantiation: behind the
// hand-over in the plain kernel, inside the loop over the two phases in the LONG one).  Context: b_from, b_to, b_after0 (the bank behind bank 0: 1, or more where the first banks were replayed ahead), eend, xs, the err_* lambdas.
      // (tried: bank by bank statically, the banks with a leaf's end noted for a second loop: no search for the bank's code, but the
      //  register allocator moves parts of the stash around between the banks' codes -- error pass 13 % slower)
#pragma nounroll
        for (unsigned int b = b_from; b < b_to; b = (b == 0u ? b_after0 : b + 1u)) {
          const unsigned int kb0 = b * (unsigned int)RG_ROW;
          // a block in which no leaf ends, and not the one with step 0: straight from the bank's registers under ONE test (one
          // copy of the 16 steps per bank: 5.5 instructions a step); else through T[] with a test per step
          if (b != 0u && __all(eend >= kb0 + 16u || eend <= kb0)) {
            if (eend > kb0) {
              auto group = [&](auto g_tag) {
                constexpr int g = decltype(g_tag)::value;
                rg_static_for<(g == 0 ? 1 : 4 * g), (4 * g + 4 < SBLK ? 4 * g + 4 : SBLK)>([&](auto i_tag) {
                  constexpr int i = decltype(i_tag)::value;
                  if (b == (unsigned int)i) {
#pragma unroll
                    for (int qq = 0; qq < RG_ROW; qq += 2) {
                      if constexpr (F32) err_pair_f(xs[i * RG_ROW + qq], (unsigned int)(i * RG_ROW + qq), xs[i * RG_ROW + qq + 1], (unsigned int)(i * RG_ROW + qq + 1));
                      else if constexpr (W2) err_pair_w(xs[i * RG_ROW + qq], (unsigned int)(i * RG_ROW + qq), xs[i * RG_ROW + qq + 1], (unsigned int)(i * RG_ROW + qq + 1));
                      else err_pair(xs[i * RG_ROW + qq], (unsigned int)(i * RG_ROW + qq), xs[i * RG_ROW + qq + 1], (unsigned int)(i * RG_ROW + qq + 1));
                    }
                    asm volatile("; stash bank %0" ::"n"(i));
                  }
                });
              };
              rg_static_for<0, (SBLK + 3) / 4>([&](auto g_tag) {
                constexpr int g = decltype(g_tag)::value;
                if (b >= (unsigned int)(4 * g) && b < (unsigned int)(4 * g + 4)) group(g_tag);
              });
            }
            continue;
          }
          XT T[RG_ROW];
          {
            auto group = [&](auto g_tag) {
              constexpr int g = decltype(g_tag)::value;
              rg_static_for<4 * g, (4 * g + 4 < SBLK ? 4 * g + 4 : SBLK)>([&](auto i_tag) {
                constexpr int i = decltype(i_tag)::value;
                if (b == (unsigned int)i) {
#pragma unroll
                  for (int qq = 0; qq < RG_ROW; qq++) T[qq] = xs[i * RG_ROW + qq];
                  asm volatile("; stash bank %0" ::"n"(i));
                }
              });
            };
            rg_static_for<0, (SBLK + 3) / 4>([&](auto g_tag) {
              constexpr int g = decltype(g_tag)::value;
              if (b >= (unsigned int)(4 * g) && b < (unsigned int)(4 * g + 4)) group(g_tag);
            });
          }
#pragma unroll
          for (int qq = 0; qq < RG_ROW; qq++) {
            const unsigned int k = kb0 + (unsigned int)qq;
            if (k < eend && k != 0u) {                                      // (step 0: above)
              if constexpr (F32) err_step_f(T[qq], k); else if constexpr (W2) err_step_w(T[qq], k); else err_step(T[qq], k);
            }
          }
        }

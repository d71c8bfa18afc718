// The short form's F5 of k_spline_scan (rmi_scan.hip.h includes this text where the leaf ends are run: a text and not a lambda -- through one, the very
// same body left 9 spilled registers in a kernel that had none, and a spilled register's reload waits for the next tile's loads).
// The leaf ends (two_layer.rs:185-197, 226-259) of the slots [0, ends_count) in 32 bits -- every index and every clamped prediction is below 2^32 --, rows,
// aggregates; then the empty leaves in front of the starts.  Lane l: slot l.  Context: ends_count, ends_A2 (the global index of the tile image's first key:
// not batched, the containers' end keys are read from the image), the slot tables, the aggregates' accumulators, the kernel's lambdas.
  {
    wave_sync();
    const sc_kargp cp = cold();
    const ScanOut out = cold_out(cp);
    const int npeers = SC_ARG(cp, int, peers.n);
    auto store_leaf = [&](unsigned int j, unsigned int s_, double a, double b2, unsigned int final_err, unsigned int cnt_j) {
      out.leaf_start[j] = (unsigned long long)s_;
      if (out.params) { out.params[2 * (size_t)j] = a; out.params[2 * (size_t)j + 1] = b2; }
      if (out.leaf_err) out.leaf_err[j] = (unsigned long long)final_err;
      if (out.leaf_count) out.leaf_count[j] = (unsigned long long)cnt_j;
      double* rp = reinterpret_cast<double*>(out.rows + (size_t)j * 24);
      rp[0] = a; rp[1] = b2;
      *reinterpret_cast<unsigned long long*>(out.rows + (size_t)j * 24 + 16) = (unsigned long long)final_err;
      for (int pi = 0; pi < npeers; pi++) {                                // (wave-uniform trip count)
        unsigned char* const pt = cold_peer(cp, pi);
        double* pr = reinterpret_cast<double*>(pt + (size_t)j * 24);
        pr[0] = a; pr[1] = b2;
        *reinterpret_cast<unsigned long long*>(pt + (size_t)j * 24 + 16) = (unsigned long long)final_err;
      }
    };
    if ((unsigned int)lane < ends_count) {
      const unsigned int q_s = r_s[lane], q_e = r_end[lane], q_tw = r_t[lane];
      const unsigned int q_t = FB ? (q_tw & 0x1FFFFFFFu) : q_tw;
      K k_lo, k_hi;
      if constexpr (FBK) { k_lo = bits_to_key<K>((B)r_klo[lane]); k_hi = bits_to_key<K>((B)r_khi[lane]); }
      else if constexpr (FB) { k_lo = keys[q_s - 1u]; k_hi = keys[q_e]; }   // (n < 2^32: the indices ARE the global ones; an ordinary tile's containers lie inside the launch)
      else {
        const int d_lo = (int)(q_s - 1u - ends_A2) * DW, d_hi = (int)(q_e - ends_A2) * DW;
        k_lo = bits_to_key<K>(bits_at(trow0 + d_lo + 4 * (d_lo >> 5)));
        if (FAR == 0 || (int)(q_e - ends_A2) < BTILE + EXTN) k_hi = bits_to_key<K>(bits_at(trow0 + d_hi + 4 * (d_hi >> 5)));
        else k_hi = keys[q_e];                                              // (FAR: the open leaf's end behind the look-ahead)
      }
      const double ma = m_ab[2 * (lane + 1)], mb = m_ab[2 * (lane + 1) + 1];
      const unsigned int curr = m_err[lane + 1], ru = m_run[lane + 1];
      const unsigned int up = min(sg_cvt_u32(__builtin_fma(mb, KeyTraits<K>::as_float(KeyTraits<K>::minus_eps(k_hi)), ma)), n32);   // lower_bound_correction.rs:47-49
      const unsigned int upper = sg_absdiff(up, min(q_e + 1u, n32));                                                              // two_layer.rs:229-235
      const unsigned int lw = min(sg_cvt_u32(__builtin_fma(mb, KeyTraits<K>::as_float(KeyTraits<K>::plus_eps(k_lo)), ma)), n32);    // lower_bound_correction.rs:62-63
      const unsigned int lower = sg_absdiff(lw, q_t == 0u ? q_e : q_s);                                                          // two_layer.rs:237-247
      const unsigned int final_err = max(max(curr, upper), lower) + (ru > 1u ? ru : 1u);                                       // :250-251 (a leaf with keys owns a recorded run)
      const unsigned int cnt_j = q_e - q_s;
      if (!(RMI_SC_DIAG & 2)) store_leaf(q_t, q_s, ma, mb, final_err, cnt_j);
      // the terms of the aggregates (two_layer.rs:267-287), as ScAgg::add forms them
      auto agg_max = [&](unsigned int j, unsigned int e) { if (e > amx || (e == amx && j > ami)) { amx = e; ami = j; } };   // max_by_key: the LAST maximum
      agg_max(q_t, final_err);
      if (!(RMI_SC_DIAG & 1)) {
        const unsigned long long ts = (unsigned long long)cnt_j * (unsigned long long)final_err;
        asum += ts;
        if (cnt_j) {
          const double v = (double)ts;
          aggp[0] += (v * v) * inv_nf;                                       // (the reference divides by n: one rounding apart, the sums are compared to 1e-9)
          aggp[1] += (double)cnt_j * sc_log2_int((double)(2ull * (unsigned long long)final_err + 2ull));
        }
      }
      // the empty leaves [g0, t) in front of this start (s == e): the constant model next_index = s (two_layer.rs:185-197), widened by 1
      const unsigned int g0 = FB ? q_t - (q_tw >> 29) : r_g0[lane];
      for (unsigned int j = g0; j < q_t; j++) {
        store_leaf(j, q_s, (double)q_s, 0.0, 1u, 0u);
        agg_max(j, 1u);
      }
    }
  }

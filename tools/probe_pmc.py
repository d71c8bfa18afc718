import collections, csv, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(sys.argv[1])):
    acc[r["Kernel_Name"][:44]][r["Counter_Name"]] += float(r["Counter_Value"])
for k, v in acc.items():
    w = v.get("SQ_WAVE_CYCLES", 1.0)
    n = max(1.0, v.get("SQ_INSTS_VALU", 0) + v.get("SQ_INSTS_SALU", 0))
    print("%-46s wait %.2f active %.2f wait_inst %.2f cycles/instr %.2f" % (k, v.get("SQ_WAIT_ANY", 0) / w, v.get("SQ_ACTIVE_INST_ANY", 0) / w, v.get("SQ_WAIT_INST_ANY", 0) / w, 4 * w / n))

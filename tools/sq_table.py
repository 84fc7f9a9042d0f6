"""Per-kernel SQ counter table from two rocprofv3 --pmc passes (tools/measure_sq.sh)."""
import collections
import csv
import re
import sys


def load(path):
    rows = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "")
        rows[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return rows, disp


a, da = load(sys.argv[1])
b, _ = load(sys.argv[2])
print("# SQ counters per kernel (sums over one bench pass of 64 candidates + warm-up launches; rocprofv3 --pmc, two passes, no tracing flags,")
print("# one stream).  wait% = SQ_WAIT_ANY / SQ_WAVE_CYCLES; mfma_busy% = SQ_VALU_MFMA_BUSY_CYCLES / (32 * SQ_BUSY_CYCLES)")
print("# (calibrated: tools/mfma_peak's register-only MFMA loop, 99 % of the spec rate, reads 31.96 for that ratio); valu/wave = SQ_INSTS_VALU / SQ_WAVES.")
for k, c in sorted(a.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0))[:24]:
    d = b.get(k, {})
    wc, bc = c.get("SQ_WAVE_CYCLES", 0), c.get("SQ_BUSY_CYCLES", 0)
    lds_act = d.get("SQ_LDS_IDX_ACTIVE", 0)
    print("%-58s n=%3d wave_cyc=%.3g wait%%=%5.1f wait_inst%%=%5.1f active_inst%%=%5.1f valu/wave=%7.0f mfma_busy%%=%5.1f lds_conflict%%=%5.1f vmem_rd=%.3g vmem_wr=%.3g"
          % (k[:58], len(da[k]), wc, 100 * c.get("SQ_WAIT_ANY", 0) / wc if wc else 0, 100 * c.get("SQ_WAIT_INST_ANY", 0) / wc if wc else 0,
             100 * c.get("SQ_ACTIVE_INST_ANY", 0) / wc if wc else 0, c.get("SQ_INSTS_VALU", 0) / max(c.get("SQ_WAVES", 1), 1),
             100 * d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (32 * bc) if bc else 0, 100 * d.get("SQ_LDS_BANK_CONFLICT", 0) / lds_act if lds_act else 0,
             d.get("SQ_INSTS_VMEM_RD", 0), d.get("SQ_INSTS_VMEM_WR", 0)))

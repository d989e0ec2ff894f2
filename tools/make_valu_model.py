#!/usr/bin/env python3
"""profiles/valu_model_rNN.json from raw measurements (all taken on the MI355X box by tools/profile_round.sh):
  op_rates.txt    tools/ubench/op_rates: cycles per wave-instruction per SIMD, by instruction, from kernel wall time
  clock_probe.txt tools/ubench/clock_probe: sustained shader clock under a multiply-bound kernel
  <pmc dir>       rocprofv3 --pmc SQ_INSTS_VALU over `bench.py --depth 1 --no-prove --no-cpu` (k_accumulate launches)
usage: make_valu_model.py <op_rates.txt> <clock_probe.txt> <pmc dir> <entries per launch> <out.json>"""
import csv, glob, json, re, sys

op_rates, clock, pmc_dir, entries, out = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4]), sys.argv[5]
tag = sys.argv[6] if len(sys.argv) > 6 else "r03"
commit = sys.argv[7] if len(sys.argv) > 7 else "unknown"
cyc = None
for line in open(op_rates):
    if line.startswith("mad_u64 + addc pair"):
        vals = [float(x) for x in re.findall(r"\dw:\s*([0-9.]+)", line)]
        cyc = min(vals)                      # the best sustained rate over 1, 2, 4, 8 workgroups per CU
mhz = [float(m) for m in re.findall(r"shader clock (\d+) MHz", open(clock).read())]
vals = []
for f in glob.glob(pmc_dir + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_accumulate" in r["Kernel_Name"] and r["Counter_Name"] == "SQ_INSTS_VALU":
            vals.append(float(r["Counter_Value"]))
# the run also holds a small self-check MSM (bench.py --no-cpu): keep the launches of the timed workload only
vals = [v for v in vals if v >= 0.8 * max(vals)]
per_launch = sum(vals) / len(vals)
res = {"produced_at_commit": commit, "round": tag, "valu_per_bucket_addition": per_launch * 64.0 / entries,         # SQ_INSTS_VALU counts wave-instructions
       "cycles_per_valu_wave_instruction_per_simd": cyc,
       "sustained_shader_clock_ghz": sorted(mhz)[len(mhz) // 2] / 1e3,
       "provenance": {"SQ_INSTS_VALU_per_k_accumulate_launch": per_launch, "launches": len(vals), "entries_per_launch": entries,
                      "op_rates": "profiles/" + tag + "_" + op_rates.split("/")[-1] + " (row 'mad_u64 + addc pair', best of 1/2/4/8 workgroups per CU; "
                                  "the multiply is one v_mad_u64_u32 + one v_addc_co_u32 per product)",
                      "clock": "profiles/" + tag + "_" + clock.split("/")[-1] + " (median of the reported shader clocks: the first runs ramp up from idle)",
                      "note": "MI355X_MICROARCH.md: SIMD-32, a wave64 VALU instruction issues over 2 cycles at full rate; "
                              "v_mad_u64_u32 and the carry-writing adds measure at about twice that"}}
json.dump(res, open(out, "w"), indent=1)
print(res)

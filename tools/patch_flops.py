#!/usr/bin/env python3
"""See tools/patch_flops.sh.  `run <log.json>`: one warm optimize() of the bench
scene, batch log written; `report <dir>`: counters of the gn_patch_kernel
launches per template form / active patch-steps of its scales."""
import collections, csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if sys.argv[1] == "run":
    import bench
    from smvs_amd import host
    inp = bench.scene_inputs(0)
    # (no warm-up call: every launch of the process is counted, so there is one optimize())
    r = host.optimize(inp, regularization=bench.REG, num_iterations=5, min_scale=bench.SCALE,
                      want_maps=False)
    json.dump(r["log"], open(sys.argv[2], "w"))
    sys.exit(0)

d = sys.argv[2]
log = json.load(open(os.path.join(d, "log.json")))
# which form serves which scale (gn_construct.hip, launch_patch_kernel): 16
# samples per patch at scales <= 3, 64 at 4 and 5, 256 (four chunks) at 6
form_of = lambda s: "<4, 1, false>" if s <= 3 else ("<1, 1, true>" if s <= 5 else "<1, 4, false>")
aps = collections.Counter()
for e in log:
    aps[form_of(e["scale"])] += e["active_patch_steps"]
acc = collections.defaultdict(lambda: collections.Counter())
for fn in glob.glob(d + "/s*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if "gn_patch_kernel" in r["Kernel_Name"]:
            form = r["Kernel_Name"][r["Kernel_Name"].index("<"):r["Kernel_Name"].index(">") + 1]
            acc[form][r["Counter_Name"]] += float(r["Counter_Value"])
            acc[form]["launches:" + r["Counter_Name"]] += 1
print("# FP64 flops executed by gn_patch_kernel per active patch, one optimize() of the bench scene (1920x1080, 8 neighbours)")
print("# flops = (2 FMA + MUL + ADD + TRANS) x 64 lanes + MFMA_MOPS_F64 x 512 (SQ counters, summed over the launches of a form)")
out = {}
for form, c in sorted(acc.items()):
    vec = (2 * c["SQ_INSTS_VALU_FMA_F64"] + c["SQ_INSTS_VALU_MUL_F64"] + c["SQ_INSTS_VALU_ADD_F64"]
           + c["SQ_INSTS_VALU_TRANS_F64"]) * 64.0
    mfma = c["SQ_INSTS_VALU_MFMA_MOPS_F64"] * 512.0
    n = aps.get(form, 0)
    per = (vec + mfma) / max(n, 1)
    print("%-16s launches %3d  active patch-steps %7d  vector flops %.4g  MFMA flops %.4g -> %.1f flop per patch (MFMA share %.2f)"
          % (form, c["launches:SQ_WAVES"], n, vec, mfma, per, mfma / max(vec + mfma, 1)))
    out[form] = per
by_scale = {str(s): out.get(form_of(s), 0.0) for s in sorted({e["scale"] for e in log})}
json.dump(dict(flop_per_patch_by_scale=by_scale, by_form=out,
               source="tools/patch_flops.sh: rocprofv3 --pmc SQ_INSTS_VALU_{FMA,MUL,ADD,TRANS}_F64, "
                      "SQ_INSTS_VALU_MFMA_MOPS_F64 over one optimize() of the bench scene"),
          open(os.path.join(d, "patch_flops_r6.json"), "w"), indent=1)
print(json.dumps(by_scale))

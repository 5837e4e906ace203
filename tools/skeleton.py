"""Control-flow / memory-op skeleton of a kernel's assembly (labels, branches, waitcnts, memory ops, VALU counts in between):
python tools/skeleton.py wino_dw <mangled substring> [producer|all]"""
import re, subprocess, sys
tu, pat = sys.argv[1], sys.argv[2]
part = sys.argv[3] if len(sys.argv) > 3 else "producer"
subprocess.run("cd /root/repo/neuralrgbd_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -S "
               "--cuda-device-only -I ../../include %s.hip -o /tmp/%s.s 2>/dev/null" % (tu, tu), shell=True, check=True)
s = open("/tmp/%s.s" % tu).read()
name = [m for m in re.findall(r"^(_Z\w+):", s, re.M) if pat in m][0]
k = s.split(name + ":")[1].split(".Lfunc_end")[0]
out, cnt = [], {"valu": 0, "pk": 0}
def flush():
    global cnt
    if cnt["valu"] or cnt["pk"]:
        out.append("      [valu %d pk %d]" % (cnt["valu"], cnt["pk"]))
    cnt = {"valu": 0, "pk": 0}
for l in k.split("\n"):
    t = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        flush(); out.append(m.group(1)); continue
    if not t or t.startswith((";", ".")):
        continue
    op = t.split()[0]
    if op.startswith("v_mfma"):
        flush(); out.append("MFMA"); continue
    if op.startswith("v_pk"):
        cnt["pk"] += 1
    elif op.startswith("v_"):
        cnt["valu"] += 1
    elif op.startswith(("s_waitcnt", "s_barrier", "s_cbranch", "s_branch", "global_load", "global_store", "ds_", "scratch_")):
        flush(); out.append("   " + t[:70])
res = []
for o in out:
    if res and res[-1][0] == o:
        res[-1][1] += 1
    else:
        res.append([o, 1])
txt = "\n".join("%s%s" % (o, (" x%d" % n if n > 1 else "")) for o, n in res)
if part == "producer":
    txt = txt[:txt.find("MFMA")]
print(txt)

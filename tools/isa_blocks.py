"""Per basic block of each kernel in a hipcc -S file: instruction counts by class (scratch spills, MFMA, transcendental, LDS, VMEM,
barriers).  python tools/isa_blocks.py file.s [kernel-name-substring]"""
import re
import sys


def main(path, want=""):
    lines = open(path).read().split("\n")
    starts = [i for i, l in enumerate(lines) if re.match(r"^_Z\w+:", l) and "kernel" in l]
    for k, s in enumerate(starts):
        name = lines[s].split(":")[0]
        if want not in name:
            continue
        e = starts[k + 1] if k + 1 < len(starts) else len(lines)
        body = lines[s:e]
        end = [i for i, l in enumerate(body) if l.strip().startswith("s_endpgm")]
        body = body[: end[-1] + 1] if end else body
        print(name, len(body), "lines")
        cur, stats, order = "entry", {}, []
        for l in body:
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                cur = m.group(1)
            if cur not in stats:
                stats[cur] = dict(n=0, sld=0, sst=0, mfma=0, trans=0, lds=0, vmem=0, bar=0, valu=0)
                order.append(cur)
            t = l.strip()
            if not t or t.startswith((";", ".")):
                continue
            s_ = stats[cur]
            s_["n"] += 1
            op = t.split()[0]
            if op.startswith("scratch_load"): s_["sld"] += 1
            elif op.startswith("scratch_store"): s_["sst"] += 1
            elif op.startswith("v_mfma"): s_["mfma"] += 1
            elif op.startswith(("v_exp", "v_rcp", "v_rsq", "v_sqrt", "v_log")): s_["trans"] += 1
            elif op.startswith("ds_"): s_["lds"] += 1
            elif op.startswith(("global_", "buffer_", "flat_")): s_["vmem"] += 1
            elif op.startswith("s_barrier"): s_["bar"] += 1
            elif op.startswith("v_"): s_["valu"] += 1
        for b in order:
            s_ = stats[b]
            if s_["n"] >= 20 or s_["sld"] or s_["sst"]:
                print("   %-12s %s" % (b, " ".join("%s=%d" % kv for kv in s_.items() if kv[1])))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")

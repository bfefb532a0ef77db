"""profiles/traffic.json from the committed rocprofv3 counter summaries: per (workload, kernel) the fabric
bytes of one hop launch = (2 x FETCH_SIZE + WRITE_SIZE) KiB (the counters are in units of 1024 bytes:
WRITE_SIZE x 1024 is exactly the result slab on the target line) (gfx950 FETCH_SIZE correction,
MI355X_MICROARCH.md HBM section).  python tools/update_traffic.py r3 target:spmm_mix c3:spmm_mix c2:spmm_tiled:8 ...
(a third field = the number of time pieces a pass cuts a hop into: the counters are per dispatch, the table
is per hop over the whole time axis, which is what bench.py divides by its own piece count; a fourth field = the number of hops the profiled command ran, for
operators whose hop is several launches: `c4full:spmm_split:1:6`)"""
import json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def counters(path, kernel, hops=0):
    """Per-dispatch counters of the (last) kernel whose name contains ``kernel``; with ``hops`` > 0 the TOTALS of every
    such kernel divided by the number of hops the profiled command ran (operators whose hop is several launches: the
    accumulating passes of a long-row operator -- both template instances of the kernel count)."""
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"\s+kernel: (.*)", line)
        if m:
            cur = m.group(1)
            continue
        m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+total (\S+)\s+per-dispatch (\S+)", line)
        if m and cur and kernel in cur:
            if hops:
                out[m.group(1)] = out.get(m.group(1), 0.0) + float(m.group(2)) / hops
            else:
                out[m.group(1)] = float(m.group(3))
    return out


def main():
    rnd = sys.argv[1]
    path = os.path.join(ROOT, "profiles", "traffic.json")
    table = json.load(open(path))
    for key in sys.argv[2:]:
        parts = key.split(":")
        wl, kernel, pieces = parts[0], parts[1], int(parts[2]) if len(parts) > 2 else 1
        hops = int(parts[3]) if len(parts) > 3 else 0        # 4th field: hops the profiled command ran (multi-launch hops)
        key = f"{wl}:{kernel}"
        summ = os.path.join("profiles", rnd, f"{wl}_summary.txt")
        c = counters(os.path.join(ROOT, summ), kernel, hops)
        w = bench.WORKLOADS[wl]
        ei, ew = bench.build_graph(w)
        d_h = w["R"] * w["L"]
        t = min(w["T"], w.get("t_chunk", w["T"]))
        alg = bench.hop_bytes(w["N"], t, d_h, int(ei.shape[1]))
        c = {k: v * pieces for k, v in c.items()}
        b = (2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        table[key] = {"bytes_per_launch": b, "fetch_size_kb_raw": c["FETCH_SIZE"], "write_size_kb": c["WRITE_SIZE"],
                      "algorithmic_bytes": alg, "ratio_to_algorithmic": round(b / alg, 3),
                      "source": f"{summ} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `python bench.py "
                                f"--workload {wl}`; bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024, gfx950 FETCH_SIZE correction "
                                f"of MI355X_MICROARCH.md)"}
        print(key, table[key]["bytes_per_launch"], table[key]["ratio_to_algorithmic"])
    json.dump(table, open(path, "w"), indent=1)


if __name__ == "__main__":
    main()

"""One table per round from the committed files under profiles/<round>/: bench line, rocprofv3 kernel
statistics and profiles/traffic.json.  The tables in DESIGN.md 6 and profiles/README.md are this output.
python tools/summarize_round.py r3 [r2 ...]"""
import csv, glob, json, os, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = name.replace("(anonymous namespace)::", "").replace("void ", "").replace("sgp_res::", "")
    return name.split("(")[0]


def main():
    traffic = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
    for rnd in sys.argv[1:] or ["r3"]:
        d = os.path.join(ROOT, "profiles", rnd)
        print(f"## {rnd}")
        print("| workload | node-steps/s | ms/pass | hop kernel | hop ms (HIP events) | frac of 8 TB/s | fabric / algorithmic | "
              "kernels (rocprofv3 average ms x calls) | cpu_baseline (node-steps/s, spread) |")
        print("|---|---|---|---|---|---|---|---|---|")
        for path in sorted(glob.glob(os.path.join(d, "bench_*_line*.json"))):
            w = os.path.basename(path)[len("bench_"):].split("_line")[0]
            try:
                b = json.load(open(path))
            except ValueError:
                continue
            r = b.get("roofline", {})
            kern = r.get("kernel", "?")
            t = traffic.get(f"{w}:{kern}")
            ratio = f"{t['ratio_to_algorithmic']:.2f}" if t and rnd in t["source"] else "-"
            ks = os.path.join(d, f"{w}_kernel_stats.csv")
            kernels = []
            if os.path.exists(ks) and "_line." in os.path.basename(path):
                for row in list(csv.DictReader(open(ks)))[:6]:
                    n = short(row["Name"])
                    if n.startswith("at::") or n.startswith("__amd") or float(row["Percentage"]) < 0.4:
                        continue
                    kernels.append(f"{n} {float(row['AverageNs']) / 1e6:.2f} x {row['Calls']}")
            c = b.get("cpu_baseline") or {}
            cpu = f"{c['value']:.3g} ({c.get('cores')} thr" + (f", spread {c['spread']:.0%}" if "spread" in c else "") + ")" if c else "-"
            print(f"| {w} ({os.path.basename(path)}) | {b['value']:.3g} | {b['ms_per_step']:.1f} | {kern} | "
                  f"{r.get('ms_per_launch', 0):.2f} | {r.get('frac', 0):.3f} | {ratio} | {'; '.join(kernels) or '-'} | {cpu} |")
        print()


if __name__ == "__main__":
    main()

"""A/B timing of library variants of the SAMPLER on the same box, interleaved: python scripts/ab_rj.py libA.so libB.so [...]
(AB_SIZES = chains per block, default "1024,8192"; AB_ROUNDS = repeats, default 2; ten-frequency system, reference-Jacobian mode,
gbp_rj_run's own choice of driver).  Each (variant, size) runs in its own process through scripts/bench_rj_parts.py."""
import os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:]
sizes = [int(v) for v in os.environ.get("AB_SIZES", "1024,8192").split(",")]
rounds = int(os.environ.get("AB_ROUNDS", "2"))
res = {(l, B): [] for l in libs for B in sizes}
host = {}
for rnd in range(rounds):
    for B in sizes:
        for l in libs:
            env = dict(os.environ, GBP_SYSTEM=os.environ.get("GBP_SYSTEM", "syn10"), GBP_MODES=os.environ.get("AB_MODE", "0"))
            if l != "product":
                env["GBP_AB_LIB"] = os.path.abspath(l)
            out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "bench_rj_parts.py"), str(B), str(2000 if B <= 2048 else 1000)],
                                 capture_output=True, text=True, env=env)
            m = re.findall(r"-> ([0-9.]+) M chain-it/s", out.stdout)
            h = re.findall(r"host issue ([0-9.]+) ms/it, total ([0-9.]+) ms/it", out.stdout)
            if not m:
                print(out.stdout[-1500:], out.stderr[-3000:]); raise SystemExit(1)
            res[(l, B)].append(float(m[-1]))
            host.setdefault((l, B), []).append(h[-1] if h else ("?", "?"))
for B in sizes:
    for l in libs:
        v = res[(l, B)]
        print("%-28s B=%6d  M chain-it/s: %s  -> best %.2f   (host issue / total ms per iteration: %s)" % (
            os.path.basename(l), B, " ".join("%.2f" % x for x in v), max(v), " ".join("%s/%s" % hh for hh in host.get((l, B), []))), flush=True)

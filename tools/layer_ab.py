"""Interleaved per-layer A/B of several builds of libglass.so in ONE process on one box (cdna_hip_programming.md section 5.4 rule 24):
    python tools/layer_ab.py --rounds 7 --rows 'gldsp|conv_s2' base=clip_glass_amd/libglass.so r5=tools/lib/libglass_r5.so ...
Every round runs one fully instrumented single-stream pass (P = 64, headline geometry) per library, in turn; the table is the MEDIAN
microseconds per launch of every matching layer tag and of their sum, plus the median whole pass."""
import argparse, os, re, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--rows", default=".")
    ap.add_argument("--pop", type=int, default=64)
    ap.add_argument("--mode", type=int, default=0, help="stream mode of the instrumented passes (0 one stream; 2 CLIP beside D: use ONE library per process)")
    ap.add_argument("--timed", type=int, default=0, help="also: N un-instrumented passes per library and round in the engine's default two-stream mode, wall ms per pass")
    ap.add_argument("libs", nargs="+")
    a = ap.parse_args()
    from clip_glass_amd import synth, engine as E
    ch, lat, mp, clip = synth.FFHQ_CHANNELS, 512, 8, (768, 12, 12, 32, 224, 512)
    sd = synth.make_state(synth.stylegan2_g_spec(ch, lat, mp), 0)
    sd.update(synth.make_state(synth.stylegan2_d_spec(ch), 0))
    sd.update(synth.make_state(synth.clip_visual_spec(*clip[:1], clip[1], clip[3], clip[4], clip[5]), 0))
    engs = []
    for spec in a.libs:
        name, path = spec.split("=", 1)
        E._lib = None
        os.environ["GLASS_LIB"] = os.path.abspath(path)
        eng = E.Engine(ch[::-1], latent_size=lat, mapping_layers=mp, batch_size=4, use_discriminator=True, n_obj=2, max_pop=a.pop,
                       clip=clip, noise_mode=1, noise_seed=1234)
        eng.load_state(sd)
        eng.finalize()
        eng.set_target(np.ones(clip[5], np.float32))
        eng.set_overlap(0)
        eng.evaluate(synth.latents(999, a.pop, lat))
        engs.append((name, eng))
    pat = re.compile(a.rows)
    data = {n: {} for n, _ in engs}
    for r in range(a.rounds):
        for n, eng in engs:
            eng.set_overlap(a.mode)
            eng.set_profiling(True)
            eng.evaluate(synth.latents(1000 + r, a.pop, lat), generation=r)
            tot = 0.0
            for row in eng.profile():
                us = row["total_ms"] / max(row["launches"], 1) * 1e3
                tot += row["total_ms"] * 1e3
                data[n].setdefault(row["name"], []).append(us)
            data[n].setdefault("~pass", []).append(tot)
            eng.set_profiling(False)
            eng.set_overlap(0)
    names = [n for n, _ in engs]
    if a.timed:
        import time
        wall = {n: {0: [], 2: []} for n in names}
        pops = [synth.latents(2000 + i, a.pop, lat) for i in range(a.timed)]
        for r in range(a.rounds):
            for mode in (2, 0):
                for n, eng in engs:
                    eng.set_overlap(mode)
                    eng.evaluate(pops[0], generation=50)
                    t0 = time.perf_counter()
                    for i in range(a.timed):
                        eng.evaluate(pops[i], generation=100 + i)
                    wall[n][mode].append((time.perf_counter() - t0) / a.timed * 1e3)
                    eng.set_overlap(0)
        for mode in (2, 0):
            print("%-60s" % ("wall ms per pass, stream mode %d (median / min of %d x %d)" % (mode, a.rounds, a.timed)) +
                  "".join("%10s" % ("%.2f/%.2f" % (float(np.median(wall[n][mode])), min(wall[n][mode]))) for n in names))
    rows = [k for k in data[names[0]] if pat.search(k) and k != "~pass"]
    print("%-60s" % ("median us per launch, %d interleaved rounds" % a.rounds) + "".join("%10s" % n[:9] for n in names))
    sums = {n: 0.0 for n in names}
    for k in rows:
        line = "%-60s" % k[:60]
        for n in names:
            v = float(np.median(data[n].get(k, [float("nan")])))
            sums[n] += v
            line += "%10.1f" % v
        print(line)
    print("%-60s" % "sum of the rows" + "".join("%10.1f" % sums[n] for n in names))
    print("%-60s" % "whole instrumented pass" + "".join("%10.1f" % float(np.median(data[n]["~pass"])) for n in names))


if __name__ == "__main__":
    main()

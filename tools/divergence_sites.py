#!/usr/bin/env python3
"""Where do the product's paths part from the exact ones? (round 5, VERDICT item 3: the per-site table)

The same camera paths are followed twice, bounce by bounce, at stage level: once shaded by the PRODUCT's shade stage (hipr_debug_shade of libhiprenderer.so: hardware
sin / cos / rcp / sqrt, contraction) and once by the VERIFICATION build's (which equals the oracle bit for bit); the oracle's search traces both populations (the
device's searches are bit-identical to it). Each population feeds its OWN records forward, as the renderers do. A path is followed until the two copies first differ
grossly, and that first event is classified:
  other triangle hit / hit against miss or light   the traced ray, perturbed in its last bits, lands on another primitive (an edge or a silhouette in between)
  hit accepted against refused                     coverage cut-off or back-face rule falls the other way
  shadow ray emitted or not, other light kept      the RIS reservoir keeps another candidate
  other direction sampled                          same hit, same inputs to rounding, another lobe / branch / a badly conditioned direction
Until then the copies agree to rounding; the table also gives how far the agreeing copies are apart (relative difference of throughput and direction per bounce).
usage: tools/divergence_sites.py [scene] [--accumulations 8] [--size 160x90] [--out file.json]"""
import argparse, json, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests")); sys.path.insert(0, str(ROOT / "tools"))
from verify_probe import make
from device_host_bindings import camera_paths


def main():
    p = argparse.ArgumentParser()
    p.add_argument("scene", nargs="?", default="atrium")
    p.add_argument("--accumulations", type=int, default=8)
    p.add_argument("--size", default="160x90")
    p.add_argument("--out", default=None)
    args = p.parse_args()
    w, h = (int(v) for v in args.size.split("x"))
    from bifrost3d_amd import capi
    from bifrost3d_amd.renderer import Context
    from oracle_bindings import get_oracle
    oracle = get_oracle(True)
    oracle.lib.oracle_set_f64_transcendentals(1)
    scene, bounces = make(args.scene)
    product, verify = Context(0), Context(0, arithmetic="exact")
    product.upload_scene(scene); verify.upload_scene(scene)
    search = 0 if scene.desc.triangle_count <= 64 else (3 if scene.desc.wide8_slot_count else 1)
    sites = {}
    apart = {}
    paths_total = 0
    radiance_moved = []

    def count(name, n):
        sites[name] = sites.get(name, 0) + int(n)

    for accumulation in range(1, 1 + args.accumulations):
        cam = scene.camera(w, h, accumulations=accumulation, max_bounce_count=bounces)
        rays, thr, last, hashes, accs = camera_paths(oracle, cam, w, h, accumulation)
        paths_total += len(rays)
        state = {"P": [rays.copy(), thr.copy(), last.copy()], "V": [rays.copy(), thr.copy(), last.copy()]}
        for bounce in range(bounces + 2):
            n = len(hashes)
            if n == 0:
                break
            records, hits = {}, {}
            for key, ctx in (("P", product), ("V", verify)):
                r, t, l = state[key]
                trace = r.copy(); trace[:, 7] = np.inf
                hits[key], _ = oracle.trace_closest(scene.desc, trace, skip=l, use_bvh=search, with_lights=True)
                records[key] = ctx.debug_shade(cam, r, t, hits[key], l, hashes, accs)
            P, V = records["P"], records["V"]
            idP, idV = hits["P"][:, 3].view(np.uint32), hits["V"][:, 3].view(np.uint32)
            fP, fV = P[:, 0].view(np.uint32), V[:, 0].view(np.uint32)
            surface = lambda i: (i != 0xFFFFFFFF) & ((i & 0x80000000) == 0)
            diverged = np.zeros(n, bool)
            other_primitive = (idP != idV)
            kinds = other_primitive & (surface(idP) != surface(idV))
            count("hit against miss or light", kinds.sum()); count("other triangle hit", (other_primitive & ~kinds).sum())
            diverged |= other_primitive
            rest = ~diverged
            e = rest & (((fP ^ fV) & 4) != 0); count("hit accepted against refused", e.sum()); diverged |= e
            rest = ~diverged
            e = rest & (((fP ^ fV) & 2) != 0); count("shadow ray emitted or not", e.sum()); diverged |= e
            rest = ~diverged
            both_shadow = rest & ((fP & 2) != 0)
            e = both_shadow & (np.abs(P[:, 21:24] - V[:, 21:24]).max(axis=1) > 1e-3); count("other light candidate kept", e.sum()); diverged |= e
            rest = ~diverged
            e = rest & (((fP ^ fV) & 1) != 0); count("path ended in one copy only", e.sum()); diverged |= e
            rest = ~diverged
            on = rest & ((fP & 1) != 0)
            e = on & (np.abs(P[:, 8:11] - V[:, 8:11]).max(axis=1) > 1e-2); count("other direction sampled", e.sum()); diverged |= e
            # what the event moves: the radiance this stage adds + carries, as a crude size (the paths' later contributions differ too)
            moved = np.abs(P[diverged, 1:4] + P[diverged, 24:27] - V[diverged, 1:4] - V[diverged, 24:27]).max(axis=1) if diverged.any() else np.zeros(0)
            radiance_moved.append(moved)
            alike = on & ~diverged
            if alike.any():
                d_dir = np.abs(P[alike, 8:11] - V[alike, 8:11]).max(axis=1)
                d_thr = np.abs(P[alike, 12:15] - V[alike, 12:15]).max(axis=1) / (np.abs(V[alike, 12:15]).max(axis=1) + 1e-6)
                d_org = np.abs(P[alike, 4:7] - V[alike, 4:7]).max(axis=1)
                a = apart.setdefault(bounce, {"paths": 0, "direction": [], "throughput": [], "origin": []})
                a["paths"] += int(alike.sum()); a["direction"].append(d_dir); a["throughput"].append(d_thr); a["origin"].append(d_org)
            keep = alike      # follow on only the copies that still agree: the first event of every path is what is classified
            for key, rec in (("P", P), ("V", V)):
                state[key] = [np.ascontiguousarray(rec[keep, 4:12]), np.ascontiguousarray(rec[keep, 12:16]), np.ascontiguousarray(rec[keep, 16]).view(np.uint32)]
            hashes, accs = hashes[keep], accs[keep]
    total_events = sum(sites.values())
    report = {"scene": args.scene, "frame": [w, h], "accumulations": args.accumulations, "paths": paths_total, "first_events": sites, "paths_that_part": total_events,
              "share_of_paths": total_events / max(1, paths_total), "agreeing_copies_by_bounce": {}}
    print(f"{args.scene} {w}x{h}, {args.accumulations} accumulations: {paths_total} paths, {total_events} part before they end ({100.0 * total_events / paths_total:.3f} %)")
    for name, v in sorted(sites.items(), key=lambda kv: -kv[1]):
        print(f"    {name:34s} {v:7d}   {100.0 * v / max(1, total_events):5.1f} % of the events   {v / paths_total:.2e} per path")
    for bounce, a in sorted(apart.items()):
        row = {"paths": a["paths"]}
        for field in ("direction", "throughput", "origin"):
            x = np.concatenate(a[field])
            row[field] = {"median": float(np.median(x)), "p99": float(np.quantile(x, 0.99)), "max": float(x.max())}
        report["agreeing_copies_by_bounce"][str(bounce)] = row
        print(f"    after bounce {bounce}: {a['paths']:7d} agreeing copies; next direction apart median {row['direction']['median']:.1e} / 99 % {row['direction']['p99']:.1e} / max {row['direction']['max']:.1e}; "
              f"origin {row['origin']['median']:.1e} / {row['origin']['p99']:.1e} / {row['origin']['max']:.1e}; throughput (relative) {row['throughput']['median']:.1e} / {row['throughput']['p99']:.1e} / {row['throughput']['max']:.1e}")
    moved = np.concatenate(radiance_moved) if radiance_moved else np.zeros(0)
    if len(moved):
        report["radiance_moved_at_the_event"] = {"median": float(np.median(moved)), "p90": float(np.quantile(moved, 0.9)), "max": float(moved.max())}
        print(f"    radiance the parting stage itself moves: median {np.median(moved):.2e}, 90 % {np.quantile(moved, 0.9):.2e}, max {moved.max():.2e}")
    if args.out:
        Path(args.out).parent.mkdir(parents=True, exist_ok=True)
        Path(args.out).write_text(json.dumps(report, indent=1) + "\n")
    product.close(); verify.close()


if __name__ == "__main__":
    main()

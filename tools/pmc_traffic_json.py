"""profiles/<round>_<name>_pmc_traffic.json from the two PMC summaries a profile script has just written (rocpd_pmc_by_kernel.py: FETCH_SIZE and WRITE_SIZE in KB per
dispatch, by kernel and grid): the per-dispatch MAXIMUM of the kernel's largest grid = the level-0 launch.  bench.py reads the JSON (counters cannot be read from
inside the measured process); writing it from the text files keeps the two in step (round 5's JSONs were a run behind their text files).

    python tools/pmc_traffic_json.py <fetch.txt> <write.txt> <kernel substring> <out.json> <kernel label> [sum]
    sum: every kernel of the file (the whole ORB pipeline): the means per dispatch x dispatches per batch are added up instead
"""
import json, sys


def rows(path, pat):
    out = []
    for line in open(path):
        if line.startswith("kernel") or not line.strip():
            continue
        name, rest = line[:44], line[44:].split()
        if pat in name or pat in line:
            out.append((line[:44].strip(), int(rest[0]), rest[1], int(rest[2]), float(rest[3]), float(rest[4]), float(rest[5])))      # kernel, grid, counter, n, mean, min, max
    return out


def main():
    fetch, write, pat, dst, label = sys.argv[1:6]
    fr, wr = rows(fetch, pat), rows(write, pat)
    if not fr or not wr:
        raise SystemExit("no rows for %r in %s / %s" % (pat, fetch, write))
    if len(sys.argv) > 6 and sys.argv[6] == "sum":          # the whole pipeline: every kernel dispatched once per batch (n = the number of batches of the run), copies left out
        def total(rs):
            nb = max(r[3] for r in rs)
            return sum(r[4] for r in rs if r[3] == nb and "copyBuffer" not in r[0]), {("%s grid %d" % (r[0], r[1])): round(r[4]/1024.0, 2) for r in rs if r[3] == nb and "copyBuffer" not in r[0]}
        f, fk = total(fr); w, wk = total(wr)
        json.dump({"kernel": label, "fetch_size_kb_raw_max": f, "write_size_kb_raw_max": w,
                   "correction": "guide (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request on gfx950 -> x2; WRITE_SIZE taken as reported; narrow gather "
                                 "accesses are uncalibrated (orientation, descriptors).  Sum over the pipeline's kernels of the mean per dispatch",
                   "source": "%s, %s" % (fetch, write), "per_kernel_fetch_mb_raw": fk, "per_kernel_write_mb_raw": wk}, open(dst, "w"), indent=1)
        print(dst, f, w)
        return
    g = max(r[1] for r in fr)
    f = max(r[6] for r in fr if r[1] == g); w = max(r[6] for r in wr if r[1] == max(x[1] for x in wr))
    json.dump({"kernel": label, "fetch_size_kb_raw_max": f, "write_size_kb_raw_max": w, "grid": g,
               "correction": "guide (MI355X_MICROARCH.md, HBM section): FETCH_SIZE counts 64 B per 128-B request on gfx950 -> x2; WRITE_SIZE taken as reported; narrow gather "
                             "accesses are uncalibrated.  The per-dispatch maxima of the largest grid = the level-0 launches (the same kernel also runs on coarser levels)",
               "source": "%s, %s" % (fetch, write)}, open(dst, "w"), indent=1)
    print(dst, f, w)


if __name__ == "__main__":
    main()

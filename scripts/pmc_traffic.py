"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass on gfx950) of scripts/one_step.py into
profiles/rNN_pmc_traffic.json, the record bench.py reads `roofline.traffic` from.  The record carries the hash of the kernel
sources it was measured on (bench.kernels_sha): bench.py ignores it once a kernel has changed.

    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 scripts/one_step.py 65536 3
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 scripts/one_step.py 65536 3
    python scripts/pmc_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write 65536 wgs profiles/r02_pmc_traffic.json

Correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE (KB) x 2 for 16-byte-per-lane streaming reads on gfx950; WRITE_SIZE (KB) is
exact for 16-byte stores and float atomics."""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernels_sha  # noqa: E402


def means(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter and "pmt_" in r["Kernel_Name"]:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def short(name):
    for key in ("pmt_backward_kernel", "pmt_forward_kernel<true", "pmt_forward_kernel<false", "pmt_cnn3_backward_kernel", "pmt_cnn3_forward_bf_kernel", "pmt_cnn3_forward_kernel", "pmt_rows_forward_kernel", "pmt_rows_backward_kernel"):
        if key in name:
            return {"pmt_forward_kernel<true": "pmt_forward_kernel<train>", "pmt_forward_kernel<false": "pmt_forward_kernel"}.get(key, key)
    return None


SQ_COUNTERS = ("SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_BUSY_CYCLES")


def main():
    fetch_dir, write_dir, batch, depth, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4], sys.argv[5]
    sq_dir = sys.argv[6] if len(sys.argv) > 6 else None  # the SQ pass of scripts/collect_profiles.sh (<tag>_pmc_a): what the matrix pipe did
    fetch, write = means(fetch_dir, "FETCH_SIZE"), means(write_dir, "WRITE_SIZE")
    sq = {c: means(sq_dir, c) for c in SQ_COUNTERS} if sq_dir else {}
    kernels = {}
    for name in sorted(set(fetch) | set(write)):
        s = short(name)
        if s is None:
            continue
        f_kb, w_kb = fetch.get(name, 0.0), write.get(name, 0.0)
        kernels[s] = {"fetch_size_kb": f_kb, "write_size_kb": w_kb, "hbm_bytes_per_launch": int(2 * f_kb * 1024 + w_kb * 1024)}
        for c in SQ_COUNTERS:  # per launch, summed over the chip as rocprofv3 reports them
            if name in sq.get(c, {}):
                kernels[s][c] = sq[c][name]
    rec = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 scripts/one_step.py "
                     f"{batch} 3 (batches packed as bench.py packs them), MI355X",
           "correction": "FETCH_SIZE (KB) x 2 for 16-byte-per-lane streaming reads on gfx950 (MI355X_MICROARCH.md, HBM section); "
                         "WRITE_SIZE (KB) exact for 16-byte stores and float atomics",
           "kernels_sha": kernels_sha(), "batch_read_sets": batch, "depth": depth, "kernels": kernels}
    with open(out, "w") as f:
        json.dump(rec, f, indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()

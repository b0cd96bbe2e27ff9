"""Worker process of pipeline.phase_regions: reads a pickled list of `phasing.run_job` jobs
(("1d" | "2d", arguments)) from stdin, writes the pickled list of results to stdout.  Started as
`python -m nanorepeat_amd._phase_worker`, so it never re-imports the caller's main module and never
touches the GPU (the package's other modules load the HIP library lazily)."""
import os
import pickle
import sys


def main():
    for var in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS", "NUMEXPR_NUM_THREADS"):
        os.environ[var] = "1"            # split_alleles.py:28-32: one thread per worker
    from nanorepeat_amd import phasing
    jobs = pickle.load(sys.stdin.buffer)
    out = [phasing.run_job(j) for j in jobs]
    sys.stdout.buffer.write(pickle.dumps(out))
    sys.stdout.buffer.flush()


if __name__ == "__main__":
    main()

"""Command line of the survey driver: ``python -m geobipy_amd options_file output_directory [...]``.

Same positional arguments and switches as the reference's ``geobipy`` command (geobipy/__init__.py:76-243).  Instead of
``--mpi`` under mpirun, launch one process per GPU with ``python -m torch.distributed.run --nproc-per-node N -m geobipy_amd
...``: each rank takes a block of soundings and rank 0 writes the results, one ``<line number>.npz`` per flight line.
"""
import argparse
import os
import shutil
import sys
import time


def parse(argv=None):
    p = argparse.ArgumentParser(prog="geobipy_amd", description="GeoBIPy's rjMCMC inversion of FDEM soundings on MI355X GPUs",
                                formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("options_file", help="User options file")
    p.add_argument("output_directory", help="Output directory for results")
    p.add_argument("--seed", dest="seed", default=None, help="Seed of the random streams (overrides the options file)")
    p.add_argument("--index", dest="index", type=int, default=None, help="Invert this data point only.")
    p.add_argument("--fiducial", dest="fiducial", type=float, default=None, help="Invert this fiducial only (needs --line).")
    p.add_argument("--line", dest="line_number", type=float, default=None, help="Invert this line (or the fiducial on this line).")
    p.add_argument("--verbose", dest="verbose", action="store_true", help="Throw warnings as errors.")
    p.add_argument("--mpi", dest="mpi", action="store_true", help="Accepted for compatibility: ranks come from torch.distributed.run.")
    p.add_argument("--data_directory", default=None, help="override data_directory in parameter file.")
    p.add_argument("--data_filename", default=None, help="override data_filename in parameter file")
    p.add_argument("--exact-jacobian", action="store_true", help="true derivative in the proposals (DESIGN.md 3.4)")
    p.add_argument("--no-hitmap", action="store_true", help="skip the conductivity-depth hit map")
    p.add_argument("--hankel-eps", type=float, default=None,
                   help="accuracy budget of the Hankel-filter abscissa window: ppm for frequency-domain data (default 1e-10, 0 = all "
                        "abscissae), relative to the inductive-limit value for time-domain data (default 1e-12)")
    p.add_argument("--no-containers", action="store_true",
                   help="skip the reference-layout results containers (<line>.h5 -- real HDF5, through h5py or the HDF5 C library -- or, where neither "
                        "exists, the stand-in <line>.results.npz + <line>.results.attrs.json with the same dataset paths); the per-line "
                        "summary files <line>.npz are always written")
    p.add_argument("--container", choices=("auto", "hdf5", "npz"), default="auto",
                   help="file type of the results containers: hdf5 (<line>.h5), npz (the stand-in), auto = hdf5 when it can be written")
    p.add_argument("--schedule", choices=("auto", "static", "dynamic", "lines"), default="auto",
                   help="lines: whole flight lines per rank, each rank writes the results containers of its own lines and only the "
                        "one-row summaries travel (one all_gather_into_tensor: the exchange that has run under RCCL); static: one "
                        "contiguous block of soundings per rank; dynamic: ranks draw chunks from a shared counter (the reference's "
                        "master / worker scheduling) and ship their posterior rows to rank 0 point to point (tested over gloo only); "
                        "auto = lines on more than one rank when every flight line is one run of rows of the data file, else static")
    p.add_argument("--chunk", type=int, default=None, help="soundings per block on the device (default: 16384 for static and lines, fewer when full-length traces would not fit the device budget; a 16th of a rank's share for dynamic)")
    p.add_argument("--traces", default="1",
                   help="per-iteration misfit / acceptance traces of the containers: 1 = the reference's arrays in full (2 n_markov_chains "
                        "columns), an integer stride > 1 or 'auto' (at most 4096 entries per sounding) keeps every stride-th entry, 0 = none")
    a = p.parse_args(argv)
    a.traces = "auto" if a.traces == "auto" else (int(a.traces) or None)
    if a.seed is not None:
        a.seed = int(a.seed)
    return a


def main(argv=None):
    a = parse(argv)
    if a.verbose:
        import warnings
        warnings.filterwarnings("error")
    assert os.path.exists(a.options_file), Exception("Cannot find input file {}".format(a.options_file))
    assert os.path.isdir(a.output_directory), Exception("Make sure the output directory exists {}".format(a.output_directory))
    import torch
    import torch.distributed as dist
    from . import survey
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", 0), ("WORLD_SIZE", 1), ("LOCAL_RANK", 0)))
    torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    if world > 1 and not dist.is_initialized():
        dist.init_process_group(os.environ.get("GBP_BENCH_BACKEND", "nccl"), rank=rank, world_size=world)
    if rank == 0:
        print("Running geobipy_amd on {} GPU process(es)".format(world))
        print("Using user input file {}".format(a.options_file))
        print("Output files will be produced at {}".format(a.output_directory))
        shutil.copy(a.options_file, a.output_directory)            # kept with the results, like the reference does
    containers = None if a.no_containers else a.output_directory      # reference-layout containers: FdemData, TdemData, TempestData
    t0 = time.perf_counter()
    res = survey.infer(a.options_file, seed=a.seed, index=a.index, fiducial=a.fiducial, line_number=a.line_number,
                       exact_jacobian=a.exact_jacobian, hitmap=not a.no_hitmap, hankel_eps=a.hankel_eps, schedule=a.schedule, chunk=a.chunk, traces=a.traces, results_directory=containers,
                       container=None if a.container == "auto" else a.container, data_directory=a.data_directory,
                       data_filename=a.data_filename)
    if rank == 0:
        paths = res.save_lines(a.output_directory)
        done, failed = int((res["status"] == 1).sum()), int((res["status"] == 2).sum())
        print("{} soundings: {} done, {} failed to burn in; {:.1f} s; wrote {}".format(
            res["status"].size, done, failed, time.perf_counter() - t0, ", ".join(os.path.basename(q) for q in paths)))
        if containers is not None:
            from . import hdf
            kind = hdf.container_type(None if a.container == "auto" else a.container)
            print("results containers (the reference's Inference2D / Inference1D layout): " +
                  ("HDF5 files <line>.h5, written by " + str(hdf.hdf5_writer()) if kind == "hdf5" else
                   "written as the stand-in <line>.results.npz + <line>.results.attrs.json (same dataset paths, "
                   "geobipy_amd.hdf.load_npz reads them; NOT HDF5 files -- neither h5py nor a loadable HDF5 library here, or --container npz)"))
    if world > 1:
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())

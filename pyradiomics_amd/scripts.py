"""Command line front end: the argument surface, input formats (single image + mask, or a batch CSV with `Image` /
`Mask` [/ `Label`] columns) and output formats (txt / csv / json rows, per-case `features_<idx>.csv` files and NRRD
feature maps under --out-dir) of the reference's `pyradiomics` entry point (radiomics/scripts/__init__.py:24-512,
scripts/segment.py:37-108, scripts/voxel.py:37-88).

What differs is where the parallelism goes: the reference's `--jobs N` is a `multiprocessing.Pool` of N CPU workers,
one case each.  Here `--jobs N` starts N worker processes spread round-robin over the visible MI355X GPUs
(`--gpus`, default all): worker w drives GPU w % ngpu, and cases are handed out one at a time from a shared queue,
so long and short cases balance.  More workers than GPUs is useful -- a second worker on the same GPU overlaps its
host-side feature formulas with the first worker's kernels.  No collective: cases are independent."""
from __future__ import annotations

import argparse
import collections
import csv
import json
import logging
import os
import sys
import time

import numpy as np

logger = logging.getLogger("pyradiomics_amd.script")

# override types of the reference's parameter schema (radiomics/schemas/paramSchema.yaml), used by --setting key:value
_SETTING_TYPES = {
    "minimumROIDimensions": int, "minimumROISize": int, "geometryTolerance": float, "correctMask": bool,
    "additionalInfo": bool, "label": int, "label_channel": int, "binWidth": float, "binCount": int, "normalize": bool,
    "normalizeScale": float, "removeOutliers": float, "resampledPixelSpacing": [float], "interpolator": str,
    "padDistance": int, "distances": [int], "force2D": bool, "force2Ddimension": int, "resegmentRange": [float],
    "resegmentMode": str, "resegmentShape": bool, "preCrop": bool, "sigma": [float], "start_level": int, "level": int,
    "wavelet": str, "voxelArrayShift": int, "symmetricalGLCM": bool, "weightingNorm": str, "gldm_a": int,
    "kernelRadius": int, "maskedKernel": bool, "initValue": float, "voxelBatch": int,
    "deviceResident": bool, "fusedVoxel": bool, "compactGLSZM": bool,     # pyradiomics_amd additions
}


def get_parser():
    from . import __version__
    p = argparse.ArgumentParser(prog="pyradiomics_amd", usage="%(prog)s image|batch [mask] [Options]",
                                formatter_class=argparse.RawTextHelpFormatter)
    g = p.add_argument_group("Input")
    g.add_argument("input", metavar="{Image,Batch}FILE", help="Image file (single mode) or CSV batch file (batch mode)")
    g.add_argument("mask", nargs="?", metavar="MaskFILE", default=None, help="Mask file (single mode only)")
    g.add_argument("--param", "-p", metavar="FILE", default=None, help="Parameter file (.yml/.yaml or .json)")
    g.add_argument("--setting", "-s", metavar='"SETTING_NAME:VALUE"', action="append", default=[], type=str,
                   help="Override of a setting; may be repeated")
    g.add_argument("--jobs", "-j", metavar="N", type=int, default=1,
                   help="(Batch mode) number of worker processes, spread round-robin over the GPUs")
    g.add_argument("--gpus", metavar="LIST", default=None,
                   help="Comma separated device indices to use (default: every visible GPU)")
    g.add_argument("--validate", action="store_true", help="Only check that the input files exist")
    o = p.add_argument_group("Output")
    o.add_argument("--out", "-o", metavar="FILE", type=argparse.FileType("a"), default=sys.stdout,
                   help="File to append output to")
    o.add_argument("--out-dir", "-od", type=str, default=None,
                   help="segment mode: one features_<case>.csv per case (re-used on restart);\n"
                        "voxel mode: directory for the NRRD feature maps (default: cwd)")
    o.add_argument("--mode", "-m", choices=["segment", "voxel"], default="segment")
    o.add_argument("--skip-nans", action="store_true", help="Drop features whose value is NaN")
    o.add_argument("--format", "-f", choices=["csv", "json", "txt"], default="txt")
    o.add_argument("--format-path", choices=["absolute", "relative", "basename"], default="absolute")
    o.add_argument("--unix-path", "-up", action="store_true")
    lg = p.add_argument_group("Logging")
    lg.add_argument("--logging-level", metavar="LEVEL", default="WARNING",
                    choices=["NOTSET", "DEBUG", "INFO", "WARNING", "ERROR", "CRITICAL"])
    lg.add_argument("--log-file", metavar="FILE", default=None)
    lg.add_argument("--verbosity", "-v", nargs="?", default=3, const=4, type=int, choices=[1, 2, 3, 4, 5])
    p.add_argument("--label", "-l", metavar="N", default=None, type=int, help="(DEPRECATED) label value in the mask")
    p.add_argument("--version", action="version", version="%(prog)s " + __version__)
    return p


def parse_overrides(settings, label=None):
    """["key:value", ...] -> dict, typed as the reference's parameter schema prescribes (scripts/__init__.py:513-600)"""
    def conv(v, t):
        if t is bool:
            return v == "1" or v.lower() == "true"
        return t(v)
    out = {}
    for item in settings:
        if ":" not in item:
            logger.warning('Incorrect format for override setting "%s", missing ":"', item)
            continue
        key, value = item.split(":", 1)
        if key not in _SETTING_TYPES:
            logger.warning('Did not recognize override "%s", skipping...', key)
            continue
        t = _SETTING_TYPES[key]
        try:
            out[key] = [conv(v, t[0]) for v in value.split(",")] if isinstance(t, list) else conv(value, t)
        except (TypeError, ValueError):
            logger.warning('Could not parse value "%s" for setting "%s", skipping...', value, key)
    if label is not None:
        logger.warning('Argument "label" is deprecated, use "--setting=label:N"')
        out["label"] = label
    return out


def read_cases(path, mask=None):
    """-> [(case_idx, OrderedDict(row))] from a batch CSV (columns Image, Mask required; relative paths are taken
    relative to the CSV) or a single image / mask pair (scripts/__init__.py:256-320)"""
    if path.endswith(".csv"):
        start = os.path.dirname(os.path.abspath(path))
        with open(path, newline="") as f:
            cr = csv.DictReader(f)
            for col in ("Image", "Mask"):
                if col not in (cr.fieldnames or []):
                    raise ValueError('Required column "%s" not present in input, unable to extract features...' % col)
            cases = []
            for row_idx, row in enumerate(cr, start=2):
                if not row.get("Image") or not row.get("Mask"):
                    logger.warning("Batch L%d: Missing required Image or Mask, skipping this case...", row_idx)
                    continue
                row = collections.OrderedDict(row)
                for col in ("Image", "Mask"):
                    if not os.path.isabs(row[col]):
                        row[col] = os.path.abspath(os.path.join(start, row[col]))
                cases.append(row)
        return list(enumerate(cases, start=1)), start
    if mask is None:
        raise ValueError("Input is not recognized as batch, no mask specified, cannot compute result!")
    return [(1, collections.OrderedDict([("Image", path), ("Mask", mask)]))], os.getcwd()


# ---- workers ---------------------------------------------------------------------------------------------
_WORKER = {}


def _init_worker(param, overrides, gpus, counter, log_level):
    """runs once in every worker process: pick this worker's GPU, build its extractor"""
    logging.basicConfig(level=getattr(logging, log_level), format="[%(asctime)s] %(levelname)s %(name)s: %(message)s")
    with counter.get_lock():
        w = counter.value
        counter.value += 1
    import torch
    dev = gpus[w % len(gpus)]
    torch.cuda.set_device(dev)
    from .featureextractor import RadiomicsFeatureExtractor
    _WORKER["extractor"] = RadiomicsFeatureExtractor(param, **overrides) if param else RadiomicsFeatureExtractor(**overrides)
    _WORKER["device"] = dev
    logger.info("worker %d drives GPU %d", w, dev)


def _scalar(v):
    if isinstance(v, np.ndarray) and v.ndim == 0:
        return v.item()
    return v


def extract_segment(case_idx, case, extractor, out_dir=None):
    """scripts/segment.py:37-108: features of one case; with out_dir the row is cached in features_<idx>.csv"""
    cache = os.path.join(out_dir, "features_%d.csv" % case_idx) if out_dir else None
    if cache and os.path.isfile(cache):
        with open(cache, newline="") as f:
            rd = csv.reader(f)
            logger.info("Patient %s already processed, reading results...", case_idx)
            return collections.OrderedDict(zip(next(rd), next(rd)))
    fv = collections.OrderedDict(case)
    try:
        t = time.perf_counter()
        label = case.get("Label") or None
        channel = case.get("Label_channel") or None       # scripts/segment.py:60-66: both columns override the parameter file
        fv.update((k, _scalar(v)) for k, v in extractor.execute(case["Image"], case["Mask"],
                                                                int(label) if label is not None else None,
                                                                int(channel) if channel is not None else None).items())
        logger.info("Case %s processed in %.3f s", case_idx, time.perf_counter() - t)
    except (KeyboardInterrupt, SystemExit):
        raise
    except Exception as e:        # log, keep going with the next case (segment.py:98-100)
        logger.error("Feature extraction failed! : %s", e, exc_info=True)
    if cache:
        tmp = cache + ".tmp%d" % os.getpid()
        with open(tmp, "w", newline="") as f:
            wr = csv.DictWriter(f, fieldnames=list(fv.keys()), lineterminator="\n")
            wr.writeheader()
            wr.writerow(fv)
        os.replace(tmp, cache)
    return fv


def extract_voxel(case_idx, case, extractor, out_dir=None, unix_path=False):
    """scripts/voxel.py:37-88: feature maps of one case written as Case-<idx>_<feature>.nrrd, paths in the row"""
    from .image import Image, write_nrrd
    fv = collections.OrderedDict(case)
    out_dir = out_dir or "."
    try:
        os.makedirs(out_dir, exist_ok=True)
        t = time.perf_counter()
        label = case.get("Label") or None
        channel = case.get("Label_channel") or None
        result = extractor.execute(case["Image"], case["Mask"], int(label) if label is not None else None,
                                   int(channel) if channel is not None else None, voxelBased=True)
        for k, v in result.items():
            if isinstance(v, Image):
                target = os.path.join(out_dir, "Case-%d_%s.nrrd" % (case_idx, k))
                write_nrrd(target, v)
                fv[k] = target.replace(os.path.sep, "/") if unix_path else target
            else:
                fv[k] = _scalar(v)
        logger.info("Case %s processed in %.3f s", case_idx, time.perf_counter() - t)
    except (KeyboardInterrupt, SystemExit):
        raise
    except Exception as e:
        logger.error("Feature extraction failed! %s", e, exc_info=True)
    return fv


def _run_case(job):
    case_idx, case, mode, out_dir, unix_path = job
    ex = _WORKER["extractor"]
    if mode == "segment":
        return case_idx, extract_segment(case_idx, case, ex, out_dir)
    return case_idx, extract_voxel(case_idx, case, ex, out_dir, unix_path)


def process_cases(cases, param, overrides, mode="segment", jobs=1, gpus=None, out_dir=None, unix_path=False,
                  log_level="WARNING"):
    """-> [feature row per case, input order].  jobs == 1 runs in this process on the current GPU."""
    if out_dir:
        os.makedirs(out_dir, exist_ok=True)
    jobs = max(1, min(jobs, len(cases)))
    work = [(i, c, mode, out_dir, unix_path) for i, c in cases]
    if jobs == 1:
        from .featureextractor import RadiomicsFeatureExtractor
        if gpus:                                   # --gpus with a single job: that GPU, not whichever is current
            import torch
            if torch.cuda.is_available():
                torch.cuda.set_device(int(gpus[0]))
        _WORKER["extractor"] = RadiomicsFeatureExtractor(param, **overrides) if param else RadiomicsFeatureExtractor(**overrides)
        return [_run_case(w)[1] for w in work]
    import multiprocessing as mp
    import torch
    if gpus is None:
        gpus = list(range(max(1, torch.cuda.device_count())))
    ctx = mp.get_context("spawn")               # HIP contexts do not survive fork
    counter = ctx.Value("i", 0)
    with ctx.Pool(jobs, initializer=_init_worker, initargs=(param, overrides, gpus, counter, log_level)) as pool:
        done = dict(pool.imap_unordered(_run_case, work, chunksize=1))
    return [done[i] for i, _ in cases]


def write_results(results, out, fmt="txt", skip_nans=False, format_path="absolute", unix_path=False, relative_start="."):
    """scripts/__init__.py:418-512"""
    extra = set()
    for case in results[1:]:
        extra.update(case.keys())
    extra -= set(results[0].keys())
    headers = list(results[0].keys()) + sorted(extra)
    fmt_path = {"absolute": os.path.abspath, "basename": os.path.basename,
                "relative": lambda p: os.path.relpath(p, relative_start)}[format_path]
    for idx, case in enumerate(results, start=1):
        if skip_nans:
            for k in [k for k, v in case.items() if isinstance(v, float) and np.isnan(v)]:
                del case[k]
        for col in ("Image", "Mask"):
            case[col] = fmt_path(case[col])
            if unix_path and os.path.sep != "/":
                case[col] = case[col].replace(os.path.sep, "/")
        if fmt == "csv":
            wr = csv.DictWriter(out, headers, lineterminator="\n", extrasaction="ignore")
            if idx == 1:
                wr.writeheader()
            wr.writerow(case)
        elif fmt == "txt":
            for k, v in case.items():
                out.write("Case-%d_%s: %s\n" % (idx, k, v))
    if fmt == "json":
        class Enc(json.JSONEncoder):
            def default(self, o):
                if isinstance(o, np.ndarray):
                    return o.tolist()
                if isinstance(o, (np.floating, np.integer)):
                    return o.item()
                return json.JSONEncoder.default(self, o)
        json.dump(results, out, cls=Enc, indent=2)
    out.flush()


def main(argv=None):
    args = get_parser().parse_args(argv)
    level = {1: "CRITICAL", 2: "ERROR", 3: "WARNING", 4: "INFO", 5: "DEBUG"}[args.verbosity]
    handlers = [logging.StreamHandler(sys.stderr)]
    handlers[0].setLevel(level)
    if args.log_file:
        fh = logging.FileHandler(args.log_file)
        fh.setLevel(args.logging_level)
        handlers.append(fh)
    logging.basicConfig(level=min(getattr(logging, level), getattr(logging, args.logging_level) or 100),
                        format="[%(asctime)s] %(levelname)s %(name)s: %(message)s", handlers=handlers, force=True)
    try:
        cases, start = read_cases(args.input, args.mask)
    except ValueError as e:
        logger.error("%s", e)
        return 1
    if args.validate:
        bad = 0
        if args.param is not None and not os.path.isfile(args.param):
            logger.error("Path for specified parameter file does not exist!")
        for idx, case in cases:
            for col in ("Image", "Mask"):
                if not os.path.isfile(case[col]):
                    logger.error("%s path for case (%i/%i) does not exist!", col, idx, len(cases))
                    bad += 1
        logger.info("Validation complete, errors found in %i case(s)", bad)
        return 0
    if not cases:
        logger.error("No cases to process...")
        return 1
    try:
        gpus = [int(g) for g in args.gpus.split(",")] if args.gpus else None
        results = process_cases(cases, args.param, parse_overrides(args.setting, args.label), args.mode, args.jobs,
                                gpus, args.out_dir, args.unix_path, level)
        write_results(results, args.out, args.format, args.skip_nans, args.format_path, args.unix_path, start)
    except (KeyboardInterrupt, SystemExit):
        logger.info("Cancelling Extraction")
        return -1
    except Exception:
        logger.exception("Error extracting features!")
        return 3
    return 0

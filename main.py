"""Entry script of the GPU-backed filter: builds the run from a YAML file and runs it once, as the reference's
`python main.py` does (same Config -> Simulator -> run_once() -> plot() sequence), with a few switches on top.

    python main.py                              # config.yaml in the current directory
    python main.py --config other.yaml --batch 1024 --save
"""
import argparse

from dvi_ekf_amd import Config, Simulator


def build(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    ap.add_argument("--config", default="config.yaml", help="run description (schema of the reference's config.yaml)")
    ap.add_argument("--batch", type=int, default=None, help="independent filter instances on the GPU")
    ap.add_argument("--golden-compat", action="store_true",
                    help="plumbing that reproduces the reference's saved trajectory digit for digit")
    ap.add_argument("--save", action="store_true", help="write kf_best_<traj>.txt / imu_ref_<traj>.txt next to the data")
    ap.add_argument("--no-plot", action="store_true")
    args = ap.parse_args(argv)
    cfg = Config(args.config)
    if args.batch is not None:
        cfg.batch = args.batch
    if args.golden_compat:
        cfg.golden_compat = True
    return args, Simulator(cfg)


if __name__ == "__main__":
    args, sim = build()
    sim.run_once()
    if args.save:
        sim.kf.save()
    if not args.no_plot:
        sim.plot(compact=True)

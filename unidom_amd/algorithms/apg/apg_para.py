"""`python -m unidom_amd.algorithms.apg.apg_para --env fold_cloth1_para ...` -- parameter-aware APG: per-iteration
stiffness randomisation + normalised stiffness in the observation.  Counterpart of
/root/reference/DaXBench/daxbench/algorithms/apg/apg_para.py (flags :493-565, loop :324-444)."""
from .apg import build_parser, train


def main(argv=None):
    args = build_parser(para=True).parse_args(argv)
    return train(args, para_obs=True, randomize_stiffness=True)


if __name__ == "__main__":
    main()

"""`python -m unidom_amd.algorithms.apg.apg_no_para --env fold_cloth1 ...` -- stiffness randomisation WITHOUT the
parameter in the observation.  Counterpart of /root/reference/DaXBench/daxbench/algorithms/apg/apg_no_para.py
(identical to apg_para.py except get_obs takes no eval_min_max_stiff, see `diff` in SURVEY.md)."""
from .apg import build_parser, train


def main(argv=None):
    args = build_parser(para=True).parse_args(argv)
    return train(args, para_obs=False, randomize_stiffness=True)


if __name__ == "__main__":
    main()

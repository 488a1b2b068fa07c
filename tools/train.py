#!/usr/bin/env python
"""Training entry point (reference: tools/train.py:17-40):
    python tools/train.py --config-file configs/scannet/pretrain-ponder-spunet-v1m1-synthetic.py \
        --num-gpus 1 --options save_path=exp/synthetic epoch=2 eval_epoch=2
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from ponderv2_amd.ponder.engines import (default_argument_parser, default_config_parser,  # noqa
                                         default_setup, launch)
from ponderv2_amd.ponder.engines.train import TRAINERS  # noqa: E402


def main_worker(cfg):
    cfg = default_setup(cfg)
    TRAINERS.build(dict(type=cfg.train.type, cfg=cfg)).train()


def main():
    args = default_argument_parser().parse_args()
    cfg = default_config_parser(args.config_file, args.options)
    launch(main_worker, num_gpus_per_machine=args.num_gpus, num_machines=args.num_machines,
           machine_rank=args.machine_rank, dist_url=args.dist_url, cfg=(cfg,))


if __name__ == "__main__":
    main()

from .defaults import (create_ddp_model, default_argument_parser, default_config_parser,  # noqa
                       default_setup)
from .launch import launch  # noqa: F401
from .train import TRAINERS, Trainer  # noqa: F401

"""TrainingCallback / TrainingCallbackLocation / TrainingCallbackAttributes — what FruitModel.get_training_callbacks
returns to Nerfstudio's Trainer (/root/reference/fruit_nerf/fruit_nerf.py:191-223, imports :29-30).

Inside a Nerfstudio host the host's own classes are used (the Trainer compares `location in callback.where_to_run`
with ITS enum members, so the enum must be the host's).  Without nerfstudio (this repo's tests, bench.py) the classes
below stand in: same constructor arguments, same `run_callback` / `run_callback_at_location` behaviour
(nerfstudio/engine/callbacks.py, 0.3.2)."""
from __future__ import annotations

from dataclasses import dataclass
from enum import Enum, auto
from typing import Callable, Dict, List, Optional, Tuple

try:  # pragma: no cover - exercised only inside a Nerfstudio installation
    from nerfstudio.engine.callbacks import (TrainingCallback, TrainingCallbackAttributes,  # noqa: F401
                                              TrainingCallbackLocation)
except ImportError:

    class TrainingCallbackLocation(Enum):
        """Where in Trainer.train_iteration the callback runs."""
        BEFORE_TRAIN_ITERATION = auto()
        AFTER_TRAIN_ITERATION = auto()

    @dataclass
    class TrainingCallbackAttributes:
        """What a Trainer offers its callbacks (optimizers, grad scaler, pipeline); FruitModel reads none of them."""
        optimizers: Optional[object] = None
        grad_scaler: Optional[object] = None
        pipeline: Optional[object] = None

    class TrainingCallback:
        """Runs `func(*args, **kwargs, step=step)` every `update_every_num_iters` steps, or at the steps in `iters`."""

        def __init__(self, where_to_run: List[TrainingCallbackLocation], func: Callable,
                     update_every_num_iters: Optional[int] = None, iters: Optional[Tuple[int, ...]] = None,
                     args: Optional[List] = None, kwargs: Optional[Dict] = None):
            assert "step" in func.__code__.co_varnames[:func.__code__.co_argcount + func.__code__.co_kwonlyargcount], \
                f"'step: int' must be an argument in the callback function 'func': {func.__name__}"
            self.where_to_run = where_to_run
            self.update_every_num_iters = update_every_num_iters
            self.iters = iters
            self.func = func
            self.args = args if args is not None else []
            self.kwargs = kwargs if kwargs is not None else {}

        def run_callback(self, step: int) -> None:
            if self.update_every_num_iters is not None:
                if step % self.update_every_num_iters == 0:
                    self.func(*self.args, **self.kwargs, step=step)
            elif self.iters is not None:
                if step in self.iters:
                    self.func(*self.args, **self.kwargs, step=step)

        def run_callback_at_location(self, step: int, location: TrainingCallbackLocation) -> None:
            if location in self.where_to_run:
                self.run_callback(step=step)

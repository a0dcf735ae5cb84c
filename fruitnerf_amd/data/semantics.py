"""The `metadata["semantics"]` object FruitModel requires (fruit_nerf.py:71-76): nerfstudio's
`nerfstudio.data.dataparsers.base_dataparser.Semantics`, as FruitNerfDataParser builds it
(/root/reference/fruit_nerf/data/fruitnerf_dataparser.py:251-258: classes ['apple', 'stuff'], colours [0, 255] / 255).
A Nerfstudio host passes its own Semantics instance; FruitModel only reads `.colors`, so any object with the same
attributes is accepted."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, List

import torch


@dataclass
class Semantics:
    filenames: List = field(default_factory=list)
    classes: List[str] = field(default_factory=lambda: ["apple", "stuff"])
    colors: torch.Tensor = field(default_factory=lambda: torch.tensor([0.0, 255.0]) / 255.0)
    mask_classes: List[str] = field(default_factory=list)


def apple_metadata() -> Dict:
    """`train_dataset.metadata` of the reference's dataparser for a fruit scene (fruit_pipeline.py:107)."""
    return {"semantics": Semantics(mask_classes=["apple", "stuff"])}

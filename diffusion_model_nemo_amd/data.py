"""`pixel_values` adapter: the data format on the input side of the path (data/hf_vision_data.py:14-54,57-97).

The reference turns each dataset row's PIL image (column `image` or `img`) into a float32 CHW tensor in [-1, 1] under
the key `pixel_values` (random horizontal flip when training), and maps samples back to HWC uint8 / PIL for viewing.
This module does the same with numpy + torch only (torchvision is not a dependency here): same keys, same value
ranges, same layouts. Datasets are taken as objects (a `datasets.Dataset`, a list of rows, a dict of columns); nothing
is fetched by name.
"""
from typing import Callable, Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch

IMAGE_KEYS = ("image", "img")


def get_image_key(examples: dict) -> str:
    """Column that holds the images (data/hf_vision_data.py:23-36)."""
    for key in IMAGE_KEYS:
        if key in examples:
            return key
    raise ValueError(f"Could not retrieve image from the dataset. Tried {list(IMAGE_KEYS)}, which "
                     f"did not match dataset columns : {list(examples.keys())}")


def _to_chw_unit(image) -> torch.Tensor:
    """PIL image / HWC or HW uint8 array / CHW float tensor -> float32 CHW in [0, 1] (what ToTensor produces)."""
    if isinstance(image, torch.Tensor):
        t = image.float()
        return t if t.dim() == 3 else t.unsqueeze(0)
    arr = np.asarray(image)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    if arr.ndim != 3:
        raise ValueError(f"expected an image of 2 or 3 dimensions, got shape {arr.shape}")
    t = torch.from_numpy(np.array(arr, copy=True)).permute(2, 0, 1)   # PIL arrays are read-only views
    if arr.dtype == np.uint8:
        return t.float() / 255.0
    return t.float()


def to_pixel_values(image, scale: bool = True) -> torch.Tensor:
    """One image -> float32 CHW, in [-1, 1] when `scale` (t * 2 - 1, data/hf_vision_data.py:14-18,69-70)."""
    t = _to_chw_unit(image)
    return t * 2.0 - 1.0 if scale else t


def _resize(image, size: int, center_crop: bool):
    """Shorter side to `size` (bilinear), then an optional centre crop: Resize(size) [+ CenterCrop(size)]
    (data/hf_vision_data.py:57-64). Needs PIL for non-tensor inputs."""
    try:
        from PIL import Image
    except ImportError as e:  # pragma: no cover
        raise RuntimeError("resizing needs Pillow") from e
    if not isinstance(image, Image.Image):
        arr = np.asarray(image)
        image = Image.fromarray(arr.squeeze(-1) if arr.ndim == 3 and arr.shape[-1] == 1 else arr)
    w, h = image.size
    if w <= h:
        nw, nh = size, max(1, int(size * h / w))
    else:
        nw, nh = max(1, int(size * w / h)), size
    image = image.resize((nw, nh), Image.BILINEAR)
    if center_crop:
        left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
        image = image.crop((left, top, left + size, top + size))
    return image


def get_transform(image_size: int, scale: bool = True, center_crop: bool = False) -> Callable:
    """Resize -> [centre crop] -> CHW float in [0, 1] -> [t * 2 - 1] (data/hf_vision_data.py:57-74)."""
    def fn(image):
        return to_pixel_values(_resize(image, image_size, center_crop), scale=scale)
    return fn


def get_reverse_transform(inverse_scale: bool = True, uint: bool = False, as_pil: bool = True) -> Callable:
    """CHW tensor (in [-1, 1] when `inverse_scale`) -> HWC * 255 [-> uint8] [-> PIL image]
    (data/hf_vision_data.py:77-97). `as_pil=False` stops at the numpy array."""
    def fn(t: torch.Tensor):
        t = t.detach().float().cpu()
        if inverse_scale:
            t = (t + 1.0) / 2.0
        arr = (t.permute(1, 2, 0) * 255.0).numpy()
        if uint:
            arr = arr.astype(np.uint8)
        if not as_pil:
            return arr
        from PIL import Image
        a8 = arr if arr.dtype == np.uint8 else np.clip(arr, 0, 255).astype(np.uint8)
        return Image.fromarray(a8.squeeze(-1) if a8.shape[-1] == 1 else a8)
    return fn


def _apply(examples: dict, flip: bool, generator: Optional[torch.Generator]) -> dict:
    key = get_image_key(examples)
    out = []
    for image in examples[key]:
        t = to_pixel_values(image)
        if flip and bool(torch.rand((), generator=generator) < 0.5):
            t = t.flip(-1)
        out.append(t)
    examples["pixel_values"] = out
    del examples[key]
    return examples


def train_transforms(examples: dict, generator: Optional[torch.Generator] = None) -> dict:
    """Batch of rows -> `pixel_values` with a random horizontal flip (p = 0.5), image column removed
    (data/hf_vision_data.py:39-45)."""
    return _apply(examples, True, generator)


def test_transforms(examples: dict) -> dict:
    """Batch of rows -> `pixel_values`, no augmentation (data/hf_vision_data.py:48-54)."""
    return _apply(examples, False, None)


test_transforms.__test__ = False   # not a pytest test


class PixelValuesDataset(torch.utils.data.Dataset):
    """Rows of an existing dataset object with the train / test transform applied on access, the behaviour of
    `HFVisionDataset` (data/hf_vision_data.py:100-127) minus the fetch by name. `dataset` may be a `datasets.Dataset`
    (its `with_transform` is used), a sequence of row dicts, or a dict of equally long columns."""

    def __init__(self, dataset, mode: str = "train", generator: Optional[torch.Generator] = None):
        assert mode in ["train", "test"], "Mode must be in train or test"
        self.mode = mode
        self.generator = generator
        self._hf = None
        if hasattr(dataset, "with_transform"):
            fn = (lambda ex: train_transforms(ex, generator)) if mode == "train" else test_transforms
            self._hf = dataset.with_transform(fn)
            self._rows = None
        elif isinstance(dataset, dict):
            n = len(next(iter(dataset.values())))
            self._rows = [{k: v[i] for k, v in dataset.items()} for i in range(n)]
        else:
            self._rows = list(dataset)

    def __len__(self):
        return len(self._hf) if self._hf is not None else len(self._rows)

    def __getitem__(self, item):
        if self._hf is not None:
            return self._hf[item]
        row = {k: [v] for k, v in self._rows[item].items()}
        row = train_transforms(row, self.generator) if self.mode == "train" else test_transforms(row)
        return {k: v[0] for k, v in row.items()}


def collate_pixel_values(rows: Iterable[Dict]) -> Dict[str, torch.Tensor]:
    """Stack rows into the batch the model wrappers read: {"pixel_values": [B, C, H, W]} (+ "label" when present,
    models/conditional_ddpm.py:44-46)."""
    rows = list(rows)
    batch = {"pixel_values": torch.stack([r["pixel_values"] for r in rows])}
    for key in ("label", "labels"):
        if rows and key in rows[0]:
            batch[key] = torch.as_tensor([int(r[key]) for r in rows], dtype=torch.long)
    return batch

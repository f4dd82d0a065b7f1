"""Golden artefacts for the on-disk formats, produced by the REFERENCE's own code - test infrastructure,
build container only (/root/reference is not on the GPU box; the outputs under tests/golden/ are committed):

  * tests/golden/cropped_ref/  - a tiny cropped-dataset tree written by the unmodified `RandomCropComputer.__getitem__` /
    `random_crops` / `five_crops` of /root/reference/src/crop_datasets.py:14-123 (its heavy imports - hydra, Lightning, torchvision,
    the dataset class - are stubbed; torchvision's five_crop / crop are the documented ten lines), plus the source items (npz), so
    that stego_amd.data's reader AND writer are checked against files the reference produced;
  * tests/golden/ref_ckpt_manifest.json - key names, shapes and dtypes of the state_dict a Lightning checkpoint of the reference's
    LitUnsupervisedSegmenter holds (train_segmentation.py:53-106: net = DinoFeaturizer, the three probes, decoder), taken from the
    reference's own module classes (modules.py DinoFeaturizer with .cuda() and the weight download patched out, ClusterLookup) for
    ViT-S/8 and ViT-B/8, and the checkpoint's top-level layout as Lightning 1.2 writes it (the reference pins
    pytorch-lightning 1.2.x, environment.yml).
"""
import json
import os
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import ref_shim  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")


def _stub(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def load_reference_crop_datasets():
    """import /root/reference/src/crop_datasets.py byte for byte with its environment stubbed"""
    from PIL import Image
    ref_shim.load_reference_modules()
    u = sys.modules["utils"]
    # names crop_datasets.py takes from `from modules import *` (-> `from utils import *` in the reference)
    import modules as ref_modules
    for k, v in dict(join=os.path.join, Image=Image, prep_args=lambda: None, T=types.SimpleNamespace(ToTensor=lambda: None),
                     ToTargetTensor=lambda: None, os=os).items():
        setattr(ref_modules, k, v)
        setattr(u, k, v)
    ref_modules.__dict__.setdefault("__all__", None)
    if ref_modules.__dict__.get("__all__") is None:
        del ref_modules.__dict__["__all__"]

    def _get_image_size(img):                    # torchvision.transforms.functional._get_image_size for tensors: [w, h]
        return [img.shape[-1], img.shape[-2]]

    def crop(img, top, left, height, width):      # torchvision.transforms.functional.crop for tensors
        return img[..., top:top + height, left:left + width]

    def five_crop(img, size):                     # torchvision.transforms.functional.five_crop (tl, tr, bl, br, center)
        w, h = _get_image_size(img)
        ch, cw = size
        tl, tr = crop(img, 0, 0, ch, cw), crop(img, 0, w - cw, ch, cw)
        bl, br = crop(img, h - ch, 0, ch, cw), crop(img, h - ch, w - cw, ch, cw)
        top, left = int(round((h - ch) / 2.0)), int(round((w - cw) / 2.0))
        return tl, tr, bl, br, crop(img, top, left, ch, cw)

    _stub("data", ContrastiveSegDataset=object)
    _stub("hydra", main=lambda **kw: (lambda f: f))
    _stub("omegaconf", DictConfig=dict, OmegaConf=types.SimpleNamespace(to_yaml=str))
    _stub("pytorch_lightning")
    _stub("pytorch_lightning.utilities")
    _stub("pytorch_lightning.utilities.seed", seed_everything=lambda **kw: None)
    _stub("torchvision")
    _stub("torchvision.transforms")
    _stub("torchvision.transforms.functional", five_crop=five_crop, _get_image_size=_get_image_size, crop=crop)
    import importlib
    return importlib.import_module("crop_datasets")


def cropped_tree():
    cd = load_reference_crop_datasets()
    g = torch.Generator().manual_seed(11)
    items = []
    for i in range(2):
        img = torch.rand(3, 24, 32, generator=g)
        label = torch.randint(-1, 5, (24, 32), generator=g)
        items.append((img, label))
    root = os.path.join(OUT, "cropped_ref")
    for crop_type in ("five", "random"):
        fake = types.SimpleNamespace(crop_ratio=0.5)
        fake._get_size = lambda img, f=fake: cd.RandomCropComputer._get_size(f, img)
        cropper = (lambda i, x, f=fake: cd.RandomCropComputer.five_crops(f, i, x)) if crop_type == "five" else \
                  (lambda i, x, f=fake: cd.RandomCropComputer.random_crops(f, i, x))
        save = os.path.join(root, "cropped", "{}_{}_crop_{}".format("toyset", crop_type, 0.5))
        fake.img_dir = os.path.join(save, "img", "train")
        fake.label_dir = os.path.join(save, "label", "train")
        os.makedirs(fake.img_dir, exist_ok=True)
        os.makedirs(fake.label_dir, exist_ok=True)
        # what ContrastiveSegDataset hands RandomCropComputer.__getitem__: 'img' / 'label' = the extra_transform (cropper) applied
        fake.dataset = [dict(img=cropper(i, img), label=cropper(i, label)) for i, (img, label) in enumerate(items)]
        for i in range(len(items)):
            assert cd.RandomCropComputer.__getitem__(fake, i) is True                  # crop_datasets.py:112-123, unmodified
    np.savez_compressed(os.path.join(root, "source_items.npz"), **{"img%d" % i: it[0].numpy() for i, it in enumerate(items)},
                        **{"label%d" % i: it[1].numpy() for i, it in enumerate(items)})
    return root


def ckpt_manifest():
    ref = ref_shim.load_reference_modules()
    out = {}
    orig_cuda, orig_hub = torch.nn.Module.cuda, torch.hub.load_state_dict_from_url
    try:
        torch.nn.Module.cuda = lambda self, *a, **k: self
        for arch, dim, n_classes, extra in (("vit_small", 70, 27, 0), ("vit_base", 100, 27, 0)):
            holder = {}
            torch.hub.load_state_dict_from_url = lambda url, **k: holder["model"].state_dict()
            cfg = types.SimpleNamespace(dino_patch_size=8, dino_feat_type="feat", model_type=arch, pretrained_weights=None,
                                        projection_type="nonlinear", dropout=True)
            # DinoFeaturizer.__init__ builds self.model before it asks the hub for weights: hand its own state back
            orig_vit = ref.vits.__dict__[arch]

            def make(patch_size, num_classes, _o=orig_vit):
                holder["model"] = _o(patch_size=patch_size, num_classes=num_classes)
                return holder["model"]
            ref.vits.__dict__[arch] = make
            try:
                net = ref.DinoFeaturizer(dim, cfg)
            finally:
                ref.vits.__dict__[arch] = orig_vit
            sd = {"net." + k: v for k, v in net.state_dict().items()}
            sd.update({"train_cluster_probe." + k: v for k, v in ref.ClusterLookup(dim, n_classes).state_dict().items()})
            sd.update({"cluster_probe." + k: v for k, v in ref.ClusterLookup(dim, n_classes + extra).state_dict().items()})
            sd.update({"linear_probe." + k: v for k, v in torch.nn.Conv2d(dim, n_classes, (1, 1)).state_dict().items()})
            sd.update({"decoder." + k: v for k, v in torch.nn.Conv2d(dim, net.n_feats, (1, 1)).state_dict().items()})
            out[arch] = {"dim": dim, "n_classes": n_classes, "extra_clusters": extra,
                         "state_dict": {k: [list(v.shape), str(v.dtype)] for k, v in sd.items()}}
    finally:
        torch.nn.Module.cuda, torch.hub.load_state_dict_from_url = orig_cuda, orig_hub
    out["lightning_1_2_top_level_keys"] = ["epoch", "global_step", "pytorch-lightning_version", "callbacks", "optimizer_states",
                                           "lr_schedulers", "state_dict", "hparams_name", "hyper_parameters"]
    out["hyper_parameters_keys"] = ["n_classes", "cfg"]          # save_hyperparameters() of __init__(self, n_classes, cfg), :106
    with open(os.path.join(OUT, "ref_ckpt_manifest.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True)
    return {k: len(v["state_dict"]) for k, v in out.items() if isinstance(v, dict)}


if __name__ == "__main__":
    print(cropped_tree())
    print(ckpt_manifest())

"""The cropped-dataset tree of the reference, reader and writer:

    {root}/cropped/{dataset}_{five|random}_crop_{ratio}/img/{split}/{i}.jpg
                                                       /label/{split}/{i}.png

written by `src/crop_datasets.py:76-123` (five crops per source image: image i * 5 + crop number; labels stored + 1 as uint8 PNG so
that "unlabelled" -1 becomes 0) and read by `CroppedDataset` `src/data.py:370-400` (target - 1, mask = target == -1).
torchvision is not part of this image: crops and tensor conversion are done with PIL / numpy / torch directly."""
import os
import random
from os.path import join

import numpy as np
import torch
from PIL import Image
from torch.utils.data import Dataset


def crop_dir(root, dataset_name, crop_type, crop_ratio):
    return join(root, "cropped", "{}_{}_crop_{}".format(dataset_name, crop_type, crop_ratio))


def five_crop_boxes(height, width, crop_h, crop_w):
    """(top, left) of the four corners and the centre crop, torchvision.transforms.functional.five_crop's order (tl, tr, bl, br,
    centre; centre = round((size - crop) / 2))."""
    if crop_w > width or crop_h > height:
        raise ValueError("Requested crop size {} is bigger than input size {}".format((crop_h, crop_w), (height, width)))
    ct, cl = int(round((height - crop_h) / 2.0)), int(round((width - crop_w) / 2.0))
    return [(0, 0), (0, width - crop_w), (height - crop_h, 0), (height - crop_h, width - crop_w), (ct, cl)]


def random_crop_boxes(height, width, crop_h, crop_w, seed, n=5):
    """crop_datasets.py:14-57: box i of image `seed` from hash((seed, i, 0)) / hash((seed, i, 1))."""
    if crop_w > width or crop_h > height:
        raise ValueError("Requested crop size {} is bigger than input size {}".format((crop_h, crop_w), (height, width)))
    return [(hash((seed, i, 0)) % (height - crop_h), hash((seed, i, 1)) % (width - crop_w)) for i in range(n)]


def write_cropped(root, dataset_name, crop_type, crop_ratio, split, items):
    """items: iterable of (img float [3,H,W] in [0,1], label int [H,W] with -1 = unlabelled).  Writes five crops per item exactly
    as RandomCropComputer.__getitem__ does (crop_datasets.py:112-123).  Returns the number of files per directory."""
    if crop_type not in ("five", "random"):
        raise ValueError('Unknown crop type {}'.format(crop_type))
    save = crop_dir(root, dataset_name, crop_type, crop_ratio)
    img_dir, label_dir = join(save, "img", split), join(save, "label", split)
    os.makedirs(img_dir, exist_ok=True)
    os.makedirs(label_dir, exist_ok=True)
    n = 0
    for item, (img, label) in enumerate(items):
        H, W = img.shape[1], img.shape[2]
        ch, cw = int(H * crop_ratio), int(W * crop_ratio)                        # _get_size, crop_datasets.py:62-68
        boxes = five_crop_boxes(H, W, ch, cw) if crop_type == "five" else random_crop_boxes(H, W, ch, cw, item)
        for crop_num, (t, l) in enumerate(boxes):
            img_num = item * 5 + crop_num
            im = img[:, t:t + ch, l:l + cw]
            lb = label[t:t + ch, l:l + cw]
            img_arr = im.mul(255).add(0.5).clamp(0, 255).permute(1, 2, 0).to("cpu", torch.uint8).numpy()
            label_arr = (lb + 1).to("cpu", torch.uint8).numpy()
            Image.fromarray(img_arr).save(join(img_dir, "{}.jpg".format(img_num)), "JPEG")
            Image.fromarray(label_arr).save(join(label_dir, "{}.png".format(img_num)), "PNG")
            n += 1
    return n


def to_tensor(pil_img):
    """PIL RGB -> float [3,H,W] in [0,1] (torchvision's ToTensor)."""
    return torch.from_numpy(np.asarray(pil_img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div(255)


def to_target_tensor(pil_label):
    """data.py ToTargetTensor: PIL label -> int64 [1,H,W]."""
    return torch.as_tensor(np.array(pil_label), dtype=torch.int64).unsqueeze(0)


class CroppedDataset(Dataset):
    """Reference src/data.py:370-400, same constructor and return values: (image, target [H,W] with -1 = unlabelled, mask)."""

    def __init__(self, root, dataset_name, crop_type, crop_ratio, image_set, transform=to_tensor, target_transform=to_target_tensor):
        super().__init__()
        self.dataset_name = dataset_name
        self.split = image_set
        self.root = crop_dir(root, dataset_name, crop_type, crop_ratio)
        self.transform = transform
        self.target_transform = target_transform
        self.img_dir = join(self.root, "img", self.split)
        self.label_dir = join(self.root, "label", self.split)
        self.num_images = len(os.listdir(self.img_dir))
        assert self.num_images == len(os.listdir(self.label_dir))

    def __getitem__(self, index):
        image = Image.open(join(self.img_dir, "{}.jpg".format(index))).convert('RGB')
        target = Image.open(join(self.label_dir, "{}.png".format(index)))
        seed = np.random.randint(2147483647)            # the same random transform for image and target (data.py:388-394)
        random.seed(seed)
        torch.manual_seed(seed)
        image = self.transform(image)
        random.seed(seed)
        torch.manual_seed(seed)
        target = self.target_transform(target)
        target = target - 1
        mask = target == -1
        return image, target.squeeze(0), mask

    def __len__(self):
        return self.num_images

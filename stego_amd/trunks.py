"""`load_model` of the reference (src/utils.py:69-125) for the ResNet-50 trunks that `arch: feature-pyramid` cuts into
(src/train_segmentation.py:65-67, src/modules.py:121-205).  torchvision does not exist in this image and nothing can be
downloaded, so the trunk is defined here with torchvision's parameter names (a torchvision / MoCo / DenseCL / robust-ResNet
checkpoint placed in `data_dir` loads unchanged) and the reference's download step becomes a FileNotFoundError that names the
file.  Returned, like the reference's: `nn.Sequential(*list(resnet50.children())[:-1])` in eval mode - children 5, 6, 7 are
layer2 / layer3 / layer4, the 28 / 14 / 7 activations FeaturePyramidNet reads."""
import os
from os.path import join

import torch
import torch.nn as nn


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = nn.Conv2d(inplanes, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + (x if self.downsample is None else self.downsample(x)))


class ResNet50(nn.Module):
    """torchvision.models.resnet50 (same children order and state-dict keys): conv1, bn1, relu, maxpool, layer1..4, avgpool, fc."""

    def __init__(self, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(64, 3)
        self.layer2 = self._make_layer(128, 4, 2)
        self.layer3 = self._make_layer(256, 6, 2)
        self.layer4 = self._make_layer(512, 3, 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(2048, num_classes)

    def _make_layer(self, planes, blocks, stride=1):
        down = None
        if stride != 1 or self.inplanes != planes * 4:
            down = nn.Sequential(nn.Conv2d(self.inplanes, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))
        layers = [Bottleneck(self.inplanes, planes, stride, down)]
        self.inplanes = planes * 4
        layers += [Bottleneck(self.inplanes, planes) for _ in range(1, blocks)]
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        x = self.layer4(self.layer3(self.layer2(self.layer1(x))))
        return self.fc(torch.flatten(self.avgpool(x), 1))


_FILES = {"robust_resnet50": "imagenet_l2_3_0.pt", "densecl": "densecl_r50_coco_1600ep.pth",
          "mocov2": "moco_v2_800ep_pretrain.pth.tar", "resnet50": "resnet50-0676ba61.pth"}


def load_model(model_type, data_dir, allow_random_init=False):
    """reference utils.py:69-125 for the ResNet-50 family.  `allow_random_init=True` (tests, smoke runs) skips the
    checkpoint; otherwise the file the reference would download must already be in `data_dir`."""
    if model_type not in _FILES:
        raise ValueError("No model: {} found".format(model_type))          # (densenet121 / vgg11 need torchvision itself)
    model = ResNet50()
    path = join(data_dir, _FILES[model_type])
    if os.path.exists(path):
        ckpt = torch.load(path, map_location="cpu")
        if model_type == "robust_resnet50":                                  # utils.py:76-79
            sd = {name.split('model.')[1]: v for name, v in ckpt['model'].items() if 'model' in name}
            model.load_state_dict(sd)
        elif model_type == "densecl":                                        # utils.py:87-90
            model.load_state_dict(ckpt['state_dict'], strict=False)
        elif model_type == "mocov2":                                         # utils.py:101-111
            sd = ckpt['state_dict']
            for k in list(sd.keys()):
                if k.startswith('module.encoder_q') and not k.startswith('module.encoder_q.fc'):
                    sd[k[len("module.encoder_q."):]] = sd[k]
                del sd[k]
            msg = model.load_state_dict(sd, strict=False)
            assert set(msg.missing_keys) == {"fc.weight", "fc.bias"}
        else:
            model.load_state_dict(ckpt)
    elif not allow_random_init:
        raise FileNotFoundError("%s: the reference downloads this checkpoint (utils.py:69-111); there is no network here - "
                                "place it there, or pass allow_random_init=True" % path)
    model = nn.Sequential(*list(model.children())[:-1])
    model.eval()
    return model

"""ResNet-50 v1.5 (stride on the 3x3 of each bottleneck), pure torch.

Only what reference ``model.py:61-91`` touches: ``resnet50(weights=...)`` with
children in the order conv1,bn1,relu,maxpool,layer1..4,avgpool,fc, and the two
``*_Weights`` names imported at ``model.py:14``.
"""
import torch.nn as nn


class _W:
    IMAGENET1K_V1 = "IMAGENET1K_V1"


ResNet50_Weights = _W
DenseNet121_Weights = _W


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, cin, planes, stride, down):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride=stride, padding=1, bias=False)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, planes * 4, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None
        if down:
            self.downsample = nn.Sequential(
                nn.Conv2d(cin, planes * 4, 1, stride=stride, bias=False),
                nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        idt = x
        y = self.relu(self.bn1(self.conv1(x)))
        y = self.relu(self.bn2(self.conv2(y)))
        y = self.bn3(self.conv3(y))
        if self.downsample is not None:
            idt = self.downsample(x)
        return self.relu(y + idt)


class ResNet(nn.Module):
    def __init__(self, blocks=(3, 4, 6, 3)):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, stride=2, padding=3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, stride=2, padding=1)
        cin = 64
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), blocks)):
            layers = []
            for j in range(n):
                stride = 2 if (j == 0 and i > 0) else 1
                layers.append(Bottleneck(cin, planes, stride, j == 0))
                cin = planes * 4
            setattr(self, "layer%d" % (i + 1), nn.Sequential(*layers))
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc = nn.Linear(2048, 1000)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")


def resnet50(weights=None, pretrained=False, **kw):
    return ResNet()

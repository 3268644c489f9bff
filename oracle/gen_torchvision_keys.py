"""Writes tests/golden/torchvision_resnet50_state_dict.json: the key -> shape list of ``torchvision.models.resnet50().state_dict()``
(the file resnet50-0676ba61.pth the reference's encoder is built from, model.py:66-69,204-207).

TEST INFRASTRUCTURE ONLY.  torchvision is not installed offline; the list is written out from the PUBLISHED architecture (He et al.
2015, torchvision's ResNet v1.5: Bottleneck blocks [3, 4, 6, 3], widths 64 / 128 / 256 / 512, expansion 4, 7x7 stem, a 1x1 + BatchNorm
downsample branch in block 0 of every stage, a 1000-way fc) -- deliberately NOT derived from oracle/standins/torchvision, which the
fixture pins (tests/test_oracle_cpu.py).  320 entries: 53 conv weights, 53 BatchNorms x 5, fc x 2.
    python -m oracle.gen_torchvision_keys
"""
import json
import os


def resnet50_state_dict_spec():
    spec = []

    def bn(prefix, c):
        for leaf, shape in (("weight", [c]), ("bias", [c]), ("running_mean", [c]), ("running_var", [c]), ("num_batches_tracked", [])):
            spec.append((prefix + "." + leaf, shape))

    spec.append(("conv1.weight", [64, 3, 7, 7]))
    bn("bn1", 64)
    inplanes = 64
    for li, (planes, blocks) in enumerate(((64, 3), (128, 4), (256, 6), (512, 3)), start=1):
        for b in range(blocks):
            p = "layer%d.%d" % (li, b)
            spec.append((p + ".conv1.weight", [planes, inplanes, 1, 1]))
            bn(p + ".bn1", planes)
            spec.append((p + ".conv2.weight", [planes, planes, 3, 3]))
            bn(p + ".bn2", planes)
            spec.append((p + ".conv3.weight", [planes * 4, planes, 1, 1]))
            bn(p + ".bn3", planes * 4)
            if b == 0:
                spec.append((p + ".downsample.0.weight", [planes * 4, inplanes, 1, 1]))
                bn(p + ".downsample.1", planes * 4)
            inplanes = planes * 4
    spec.append(("fc.weight", [1000, 2048]))
    spec.append(("fc.bias", [1000]))
    return spec


if __name__ == "__main__":
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "torchvision_resnet50_state_dict.json")
    spec = resnet50_state_dict_spec()
    assert len(spec) == 320 and sum(1 for k, _ in spec if k.endswith("conv1.weight") or ".conv" in k or "downsample.0" in k) >= 53
    n_params = 0
    for k, s in spec:
        if "running" in k or "num_batches" in k:
            continue
        v = 1
        for d in s:
            v *= d
        n_params += v
    assert n_params == 25557032, n_params            # torchvision's documented parameter count of resnet50
    json.dump({"source": "published architecture (see the docstring of oracle/gen_torchvision_keys.py)", "parameters": n_params,
               "state_dict": [[k, s] for k, s in spec]}, open(out, "w"), indent=0)
    print("wrote", out, len(spec), "entries,", n_params, "parameters")

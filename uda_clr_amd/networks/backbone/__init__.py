from . import mobilenet


def build_backbone(backbone, output_stride, BatchNorm):
    if backbone == 'mobilenet':
        return mobilenet.MobileNetV2(output_stride, BatchNorm)
    raise NotImplementedError("backbone %r is not built yet (mobilenet only)" % (backbone,))

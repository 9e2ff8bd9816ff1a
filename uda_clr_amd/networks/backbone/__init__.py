from . import mobilenet, resnet


def build_backbone(backbone, output_stride, BatchNorm):
    if backbone == 'mobilenet':
        return mobilenet.MobileNetV2(output_stride, BatchNorm)
    if backbone == 'resnet':
        return resnet.ResNet101(output_stride, BatchNorm)
    raise NotImplementedError("backbone %r is not built (mobilenet and resnet are)" % (backbone,))

"""Oracle-backed stand-ins for the device ops a Trainer uses (TEST INFRASTRUCTURE): lets the CPU
tests drive the product Trainer classes' host logic (loop order, schedules, logging, checkpoints,
data-parallel plumbing) without a GPU.  Never imported by the product."""
import torch.nn.functional as F

from oracle import losses_ref, metrics_ref, proto_ref, step_ref


class OracleOps:
    seg_loss = staticmethod(step_ref.seg_loss)
    gen_prototype = staticmethod(proto_ref.gen_prototype)
    dice_coeff_2label = staticmethod(metrics_ref.dice_coeff_2label)
    pixel_acc = staticmethod(metrics_ref.pixel_acc)
    discriminative_loss = staticmethod(losses_ref.discriminative_loss)
    consistency_loss = staticmethod(losses_ref.consistency_loss)

    @staticmethod
    def photometric_augment(images, generator=None):
        return images * 0.9

    @staticmethod
    def gen_prototype_from_labels(target_map, feature):
        return proto_ref.gen_prototype(F.interpolate(target_map.clone(), size=feature.shape[2:], mode="nearest"), feature)

    @staticmethod
    def gen_prototype_retrify(oT_before, xt_feature, preds, features, T, stride):
        return proto_ref.gen_prototype_retrify(oT_before, xt_feature, preds, T, stride)

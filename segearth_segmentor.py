"""Drop-in entry point: ``Segmentor`` (upstream SegEarth-OV segmentor, reference segearth_segmentor.py:22-326; the class
where ``model_type='GEM'`` runs) on the MI355X HIP library.  Implementation: clip_decontamination_amd/segmentors.py."""
from clip_decontamination_amd.segmentors import MODELS, Segmentor as _Segmentor, get_cls_idx  # noqa: F401


@MODELS.register_module()
class Segmentor(_Segmentor):
    pass

"""Drop-in entry point: ``SegmentorEx`` with the reference's name, registry hook, constructor kwargs and methods
(reference segmentor.py:25-546), running on the MI355X HIP library.  Implementation:
clip_decontamination_amd/segmentors.py.  Usable with mmseg (``model=dict(type='SegmentorEx', ...)`` in configs/) or
stand-alone."""
from clip_decontamination_amd.segmentors import MODELS, SegmentorEx as _SegmentorEx, get_cls_idx  # noqa: F401


@MODELS.register_module()
class SegmentorEx(_SegmentorEx):
    pass

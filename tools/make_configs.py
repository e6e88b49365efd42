"""Emits configs/ -- the mmengine config surface of the drop-in (same keys and values as the reference's
configs/base_config.py + cfg_*.py so `eval.py --config configs/cfg_<dataset>.py` style launches keep working) and
the class-name files (one class per line, commas separate synonym queries; parsed by segmentor.get_cls_idx).

    python tools/make_configs.py

The table below is the single source of truth; the emitted files are generated artefacts.
"""
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "configs")

CLASS_SETS = {
    "chn6-cug": ["background", "road"],
    "deepglobe": ["urban", "agriculture", "rangeland", "forest", "water", "barren", "background"],
    "inria": ["background", "building"],
    "isaid": ["background", "ship", "store tank", "baseball diamond", "tennis court", "basketball court", "ground track field",
              "bridge", "large vehicle", "small vehicle", "helicopter", "swimming pool", "roundabout", "soccer ball field", "plane",
              "harbor"],
    "loveda": ["background", "building,roof,house", "road", "water", "barren", "forest", "agricultural"],
    "massachusetts_building": ["background", "building"],
    "openearthmap": ["background", "bareland,barren", "grass", "pavement", "road", "tree,forest", "water,river", "cropland",
                     "building,roof,house"],
    "potsdam": ["road,parking lot", "building", "low vegetation", "tree", "car", "clutter,background"],
    "roadval": ["background", "road"],
    "uavid": ["background", "building", "road", "car", "tree", "vegetation", "human"],
    "udd5": ["vegetation", "building", "road", "vehicle", "background"],
    "vaihingen": ["impervious surface", "building", "low vegetation", "tree", "car", "clutter"],
    "vdd": ["background", "facade", "road", "vegetation", "vehicle", "roof", "water"],
    "wbs-si": ["background", "water"],
    "whu": ["background", "building"],
    "xBD": ["background", "building"],
}

# cfg name -> (class file, model overrides, dataset type, data_root expr, img path, seg path, resize_448, dataset extras, persistent)
R = "{data_root}"
DATASETS = {
    "chn6-cug": ("chn6-cug", dict(prob_thd=0.8), "CHN6_CUGDataset", "''", "data/CHN6-CUG/val/image_cvt", "data/CHN6-CUG/val/label_cvt", True, {}, True),
    "deepglobe_road": ("roadval", dict(prob_thd=0.7), "RoadValDataset", "''", "data/GlobalRoadSet_Val/DeepGlobe_test_1530/image_cvt",
                       "data/GlobalRoadSet_Val/DeepGlobe_test_1530/label_cvt", True, {}, True),
    "inria": ("inria", dict(prob_thd=0.6), "InriaDataset", "payload/datasets/Inria", R + "/images/test", R + "/annotations/test", False, {}, True),
    "isaid": ("isaid", dict(prob_thd=0.4), "iSAIDDataset", "payload/datasets/iSAID", R + "/images/validation", R + "/annotations/validation",
              False, dict(reduce_zero_label=False), True),
    "loveda": ("loveda", dict(prob_thd=0.3), "LoveDADataset", "payload/datasets/LoveDA", R + "/images/validation", R + "/annotations/validation",
               False, dict(reduce_zero_label=True), True),
    "massachusetts_road": ("roadval", dict(prob_thd=0.7), "RoadValDataset", "''", "data/GlobalRoadSet_Val/Massachusetts_test_49/img",
                           "data/GlobalRoadSet_Val/Massachusetts_test_49/label_cvt", True, {}, True),
    "openearthmap": ("openearthmap", dict(prob_thd=0.1), "OpenEarthMapDataset", "''", "data/OpenEarthMap/img_dir/val", "data/OpenEarthMap/ann_dir/val",
                     True, dict(reduce_zero_label=False), True),
    "potsdam": ("potsdam", dict(prob_thd=0.1, bg_idx=5), "ISPRSDataset", "payload/datasets/Potsdam", R + "/images/validation",
                R + "/annotations/validation", False, {}, True),
    "spacenet_road": ("roadval", dict(prob_thd=0.7), "RoadValDataset", "''", "data/GlobalRoadSet_Val/SpaceNet_test_567/img",
                      "data/GlobalRoadSet_Val/SpaceNet_test_567/label_cvt", True, {}, True),
    "uavid": ("uavid", dict(prob_thd=0.3), "UAVidDataset", "payload/datasets/UAVid", R + "/images/validation", R + "/annotations/validation", False, {}, False),
    "udd5": ("udd5", dict(prob_thd=0.4, bg_idx=4), "UDD5Dataset", "payload/datasets/UDD/UDD5", R + "/val/src", R + "/val/gt", False, {}, False),
    "vaihingen": ("vaihingen", dict(prob_thd=0.1, bg_idx=5), "ISPRSDataset", "payload/datasets/Vaihingen", R + "/images/validation",
                  R + "/annotations/validation", False, {}, True),
    "vdd": ("vdd", dict(prob_thd=0.3), "VDDDataset", "payload/datasets/VDD", R + "/test/src", R + "/test/gt", False, {}, False),
    "wbs-si": ("wbs-si", dict(prob_thd=0.6), "WaterDataset", "''",
               "data/water-body-segmentation-in-satellite-images/WaterBodiesDatasetPreprocessed/WaterBodiesDatasetPreprocessed/Images",
               "data/water-body-segmentation-in-satellite-images/WaterBodiesDatasetPreprocessed/WaterBodiesDatasetPreprocessed/Masks_cvt",
               True, dict(ann_file="tools/dataset_converters/wbs-si_val.txt"), False),
    "whu_building": ("whu", dict(prob_thd=0.6), "WHUDataset", "payload/datasets/WHU-Building", R + "/images/test", R + "/annotations/test", False, {}, True),
    "whu_sat_II": ("whu", dict(prob_thd=0.7), "WHUDataset", "''", "data/WHU_Sat_II/Satellite_dataset_Ⅱ_East_Asia/1.cropped/test/image",
                   "data/WHU_Sat_II/Satellite_dataset_Ⅱ_East_Asia/1.cropped/test/label_cvt", True, dict(img_suffix=".tif", seg_map_suffix=".tif"), True),
    "xbd": ("xBD", dict(prob_thd=0.0), "xBDDataset", "payload/datasets/xBD", R + "/test_images_labels_targets/test/images",
            R + "/test_images_labels_targets/test/targets", False, {}, True),
}

BASE = '''# Generated by tools/make_configs.py -- model defaults shared by every dataset config.
# Keys = the constructor kwargs of segmentor.SegmentorEx (the drop-in of the reference class of the same name).
model = dict(
    type='SegmentorEx',
    clip_type='CLIP',
    vit_type='ViT-B/16',
    model_type='Experimental',
    ignore_residual=True,
    apply_sim_feat_up=True,
    cls_token_lambda=0.0,
    global_debias_factor=0.2,
    apply_outlier_suppression=True,
    outlier_suppression_cfg=dict(top_k=30),
    apply_similarity_enhancement=True,
    similarity_enhancement_cfg=dict(similarity_weight=1.0, temperature=1.0, add_self_similarity=True),
    sim_feat_up_cfg=dict(model_name='jbu_one', model_path='simfeatup_dev/weights/xclip_jbu_one_million_aid.ckpt'),
)

test_evaluator = dict(type='IoUMetric', iou_metrics=['mIoU'])
default_scope = 'mmseg'
env_cfg = dict(cudnn_benchmark=True, mp_cfg=dict(mp_start_method='fork', opencv_num_threads=0), dist_cfg=dict(backend='nccl'))
vis_backends = [dict(type='LocalVisBackend')]
visualizer = dict(type='SegLocalVisualizer', vis_backends=vis_backends, alpha=0.5, name='visualizer')
log_processor = dict(by_epoch=False)
log_level = 'INFO'
load_from = None
resume = False
test_cfg = dict(type='TestLoop')
default_hooks = dict(
    timer=dict(type='IterTimerHook'),
    logger=dict(type='LoggerHook', interval=50, log_metric_by_epoch=False),
    param_scheduler=dict(type='ParamSchedulerHook'),
    checkpoint=dict(type='CheckpointHook', by_epoch=False, interval=2000),
    sampler_seed=dict(type='DistSamplerSeedHook'),
    visualization=dict(type='SegVisualizationHook', interval=1))
'''


def emit_cfg(name, spec):
    cls, model, dtype, root, img, seg, resize, extras, persistent = spec
    lines = ["# Generated by tools/make_configs.py", "import os", "", "_base_ = './base_config.py'", ""]
    kw = ", ".join([f"name_path='./configs/cls_{cls}.txt'"] + [f"{k}={v!r}" for k, v in model.items()])
    lines.append(f"model = dict({kw})")
    lines.append(f"dataset_type = {dtype!r}")
    lines.append("data_root = ''" if root == "''" else f"data_root = os.path.abspath({root!r})")
    pipe = ["dict(type='LoadImageFromFile')"]
    if resize:
        pipe.append("dict(type='Resize', scale=(448, 448), keep_ratio=True)")     # annotations are loaded after the resize
    pipe += ["dict(type='LoadAnnotations')", "dict(type='PackSegInputs')"]
    lines.append("test_pipeline = [" + ", ".join(pipe) + "]")

    def path(p):
        return "f" + repr(p) if "{data_root}" in p else repr(p)
    ds = [f"type=dataset_type", "data_root=data_root"] + [f"{k}={v!r}" for k, v in extras.items()]
    ds.append(f"data_prefix=dict(img_path={path(img)}, seg_map_path={path(seg)})")
    ds.append("pipeline=test_pipeline")
    dl = ["batch_size=1", "num_workers=4"] + (["persistent_workers=True"] if persistent else [])
    dl += ["sampler=dict(type='DefaultSampler', shuffle=False)", "dataset=dict(" + ", ".join(ds) + ")"]
    lines.append("test_dataloader = dict(" + ", ".join(dl) + ")")
    with open(os.path.join(OUT, f"cfg_{name}.py"), "w", encoding="utf-8") as f:
        f.write("\n".join(lines) + "\n")


def main():
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "base_config.py"), "w") as f:
        f.write(BASE)
    for key, names in CLASS_SETS.items():
        with open(os.path.join(OUT, f"cls_{key}.txt"), "w") as f:
            f.write("\n".join(names))                 # no trailing newline: the last line must not yield an empty class
    for name, spec in DATASETS.items():
        emit_cfg(name, spec)
    print(f"wrote {1 + len(CLASS_SETS) + len(DATASETS)} files to {OUT}")


if __name__ == "__main__":
    main()

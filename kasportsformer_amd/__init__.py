"""kasportsformer_amd -- MI355X-native (gfx950) forward/backward path of KASportsFormer.

Public surface mirrors the reference's interface for this path:
  KASportsFormer, load_model            model/KASportsFormer.py:290, model/model_tools.py:79
  loss3, mpjpe_loss / ...               utils/loss_calc.py:6-27 (+ the train-step combination)
  FusedAdamW                            optim.AdamW as used in train_and_evaluate_sp.py:270-272
  DataParallel                          nn.DataParallel replacement: one process per GPU, RCCL all-reduce
  joint_flip, predict_flip_tta,         utils/utilities.py:128-135, train_and_evaluate_sp.py:27-149, utils/error_calc.py:5-48
  clip_metrics, Evaluator, evaluate_one_epoch
  pack_clip_directory, PackedClips,     data/preprocessor/clip_generate_sp.py:28-79 (file format), data/reader/sp_dataset.py:45-92
  DeviceClipLoader
  train_one_epoch                       train_and_evaluate_sp.py:201-243
  warmup_lr, ReduceLROnPlateau          train_and_evaluate_sp.py:273,325-329,393-397
  checkpoint_save, checkpoint_load      utils/utilities.py:110-118, train_and_evaluate_sp.py:171-176,285-301
  slice_source, split_clips,            data/reader/sp_reader.py:25-169,205-249, data/reader/wp_reader.py:25-135,159-199 (offline clip slicing)
  mysplit_clips, resample
"""
from .model import (KASportsFormer, load_model, set_single_stream, is_single_stream, set_deterministic, is_deterministic, set_fused_attention_backward,
                    is_fused_attention_backward)
from .functional import loss3
from .optim import FusedAdamW
from .parallel import DataParallel
from .data import PackedClips, DeviceClipLoader, pack_clip_directory, read_clip_file, shard_indices
from .checkpoint import checkpoint_save, checkpoint_load, strip_module_prefix, adamw_state_dict, load_adamw_state_dict
from .loop import train_one_epoch
from .schedule import warmup_lr, apply_warmup, ReduceLROnPlateau
from .evaluate import joint_flip, predict_flip_tta, clip_metrics, Evaluator, evaluate_one_epoch
from .synthetic import synthetic_clips, synthetic_test_extras, teacher_labels, teacher_clips
from .slicing import slice_source, split_clips, mysplit_clips, resample

__all__ = ["KASportsFormer", "load_model", "set_single_stream", "is_single_stream", "set_deterministic", "is_deterministic", "set_fused_attention_backward", "is_fused_attention_backward", "loss3", "FusedAdamW", "DataParallel", "joint_flip", "predict_flip_tta", "clip_metrics", "Evaluator",
           "evaluate_one_epoch", "PackedClips", "DeviceClipLoader", "pack_clip_directory", "read_clip_file", "shard_indices",
           "checkpoint_save", "checkpoint_load", "strip_module_prefix", "adamw_state_dict", "load_adamw_state_dict", "warmup_lr", "apply_warmup", "ReduceLROnPlateau", "train_one_epoch",
           "synthetic_clips", "synthetic_test_extras", "teacher_labels", "teacher_clips", "slice_source", "split_clips", "mysplit_clips", "resample"]

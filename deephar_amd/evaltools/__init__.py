"""Evaluation drivers that wrap `model.predict` -- the callers of the pose / action path in the reference's
exp/common/*_tools.py, with the same names, arguments and return values.  Host-side NumPy only; `model` is any
object with the Keras-Model surface of deephar_amd.Model (predict, outputs, input_shape, get_input_shape_at).

    from deephar_amd.evaltools import mpii_tools, h36m_tools, penn_tools, ntu_tools, generic

Training callbacks (MpiiEvalCallback, H36MEvalCallback, ...) are out of scope.
"""
from types import SimpleNamespace

from . import action, bbox, pose

mpii_tools = SimpleNamespace(refine_pred=bbox.refine_pred, absulute_pred=pose.absulute_pred,
                             eval_singleperson_pckh=pose.eval_singleperson_pckh)
h36m_tools = SimpleNamespace(eval_human36m_sc_error=pose.eval_human36m_sc_error)
penn_tools = SimpleNamespace(eval_singleclip_gt_bbox=action.eval_singleclip_gt_bbox,
                             eval_singleclip_gt_bbox_generator=action.eval_singleclip_gt_bbox_generator,
                             eval_multiclip_dataset=action.penn_eval_multiclip_dataset)
ntu_tools = SimpleNamespace(eval_singleclip_gt_bbox_generator=action.eval_singleclip_gt_bbox_generator,
                            eval_multiclip_dataset=action.ntu_eval_multiclip_dataset)
generic = SimpleNamespace(get_bbox_from_poses=bbox.get_bbox_from_poses)

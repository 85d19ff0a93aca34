"""Elimination-order policy of the reference, SWFOptimization::MyOrdering
(R/swf/swf_gnss.cpp:629-783), restated over global block ids.

Caller-side host logic: it fills the analogue of ceres::ParameterBlockOrdering that the
solver then follows bit-exactly.  `roles` names the blocks the reference reaches through
its member arrays (para_pose, para_speed_bias, f_manager.feature, ...):

    dummy            blackvalue2                      :645-654
    landmarks        f_manager.feature list order     :658-672
    speed_bias       para_speed_bias[i], frame order  :675-692
    poses            para_pose[i], frame order        :695-701
    mag_bias         para_bmg                         :704-708
    extrinsics       para_ex_Pose[i]                  :712-718
    blackvalue       &blackvalue                      :720-724
    spp_phase_biases / pr_corrections / rtk_ambiguities (slot order)   :727-756
    prior_kept       last_marg_info->keep_block_addr  :763-768
    parameter_head   ceres::internal::parameter_head  :772-779
    clocks           BUILD EXTENSION (SURVEY.md §8d note): per-epoch receiver-clock scalars,
                     appended to group 0 after the alternate speed-biases, epoch order.
"""
import numpy as np


def my_ordering(roles, is_const):
    order, group = [], []
    head = list(roles.get("parameter_head", []))
    mark = set(roles.get("prior_kept", [])) | set(head)           # :639-641

    def eligible(b):                                              # CONDITION, :643
        return b is not None and not is_const[b] and b not in mark

    def add(b, g):
        order.append(int(b)); group.append(int(g)); mark.add(b)

    # ---- group 0: dummy, landmarks, every other eligible speed-bias (+ clocks)
    d = roles.get("dummy")
    if d is not None:
        add(d, 0)                                                 # :645-654 (forced variable)
    for b in roles.get("landmarks", []):
        if eligible(b):
            add(b, 0)
    index = 0
    for b in roles.get("speed_bias", []):
        if eligible(b):
            if index % 2 == 0:
                add(b, 0)
            index += 1
    for b in roles.get("clocks", []):
        if eligible(b):
            add(b, 0)
    # ---- one group per remaining block, fixed order
    ors = 1
    for key in ("speed_bias", "poses"):
        for b in roles.get(key, []):
            if eligible(b):
                add(b, ors); ors += 1
    for key in ("mag_bias", "extrinsics", "blackvalue", "spp_phase_biases", "pr_corrections",
                "rtk_ambiguities"):
        v = roles.get(key, [])
        if v is None:
            continue
        if not isinstance(v, (list, tuple, np.ndarray)):
            v = [v]
        for b in v:
            if eligible(b):
                add(b, ors); ors += 1
    # ---- prior's kept blocks, then parameter_head (:760-779)
    mark = set(head)
    placed = set(order)
    for b in roles.get("prior_kept", []):
        if not is_const[b] and b not in mark:
            if b in placed:
                raise ValueError("prior block ordered twice")
            order.append(int(b)); group.append(ors); ors += 1
    n_tail = 0
    for b in head:
        if not is_const[b]:
            order.append(int(b)); group.append(ors); ors += 1; n_tail += 1
    return np.array(order, np.int32), np.array(group, np.int32), n_tail

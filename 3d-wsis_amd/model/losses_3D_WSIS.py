"""``MultiTaskLoss`` of 3D-WSIS (caller side of the hot path) -- same constructor, forward
signature, ``loss_inp`` keys and returned ``(loss, loss_out)`` as the reference's
modules/model/losses_3D_WSIS.py:13-253; pinned against the imported reference by tests/golden/loss_golden.npz.

Pure torch; ``device`` follows the inputs instead of the reference's hard-coded 'cuda' attribute (:33).

Two evaluations of the same formulas: the reference's shape (boolean-mask indexing, ``torch.unique`` -- every one of
them a device->host sync, 12+ per step) and the default MASKED one, which keeps every intermediate at its full,
host-known shape (rows that the reference drops get weight 0; per-instance means come from a dense same-instance
matrix instead of ``unique``), so the host never waits for the GPU inside the loss and can keep issuing the backward
pass while the forward is still running.  ``WSIS_LOSS_INDEXED=1`` selects the reference-shaped evaluation; both are
checked against the golden vectors."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F


class _NullLogger(object):
    def info(self, *a, **k):
        pass


class MultiTaskLoss(nn.Module):
    def __init__(self, logger, param_loss, param_model):
        super().__init__()
        self.logger = logger if logger is not None else _NullLogger()
        self.ignore_label = param_loss.ignore_label
        self.supervise_instance_size = param_loss.supervise_instance_size
        self.joint_training_epoch = param_loss.joint_training_epoch
        self.semantic_dice = param_loss.semantic_dice
        self.semantic_class_num = param_model.classes
        self.discriminative_feature_dim = 7
        self.delta_v = 0.1
        self.delta_d = 1.5
        self.param_var = 1.
        self.param_dist = 1.
        self.param_reg = 0.001
        self.supervise_sp_offset = getattr(param_loss, "supervise_sp_offset", True)
        self.log_values = getattr(param_loss, "log_values", False)   # the reference logs (= syncs) every term
        self.semantic_criterion = nn.CrossEntropyLoss(ignore_index=self.ignore_label)
        self.occupany_L1loss = nn.L1Loss()
        self.instance_size_L1loss = nn.L1Loss()
        self.superpoint_semantic_criterion = nn.CrossEntropyLoss(ignore_index=self.ignore_label)

    def _log(self, name, value):
        if self.log_values:
            self.logger.info("{}: {:.4f}".format(name, value))

    def forward(self, loss_inp, epoch):
        loss_out = {}
        semantic_labels, instance_labels = loss_inp["point_labels"]
        semantic_scores = loss_inp["semantic_scores"]
        indexed = os.environ.get("WSIS_LOSS_INDEXED", "0") == "1"
        fused = (not indexed and self.semantic_dice and semantic_scores.is_cuda and semantic_scores.shape[1] <= 32
                 and os.environ.get("WSIS_FUSE_SEM_LOSS", "1") != "0")
        if fused:       # CE + dice in two passes over [N, C] (csrc/loss.hip) instead of ~40 torch launches
            import wsis_ops
            # on the point-level head's branch stream (backbone_3D_WSIS.py): the two passes over [N, C] and their backward
            # run beside the superpoint terms instead of in front of them
            side = wsis_ops.branch_stream(semantic_scores.device, 1)
            if side is not None and os.environ.get("WSIS_BRANCH_LOSS", "0") != "0":      # (measured: within the noise)
                main = torch.cuda.current_stream()
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    semantic_loss, n_kept = wsis_ops.semantic_point_loss(semantic_scores, semantic_labels,
                                                                         self.ignore_label)
                semantic_scores.record_stream(side)
                semantic_labels.record_stream(side)
                loss_join = (main, side, (semantic_loss, n_kept))
            else:
                semantic_loss, n_kept = wsis_ops.semantic_point_loss(semantic_scores, semantic_labels, self.ignore_label)
                loss_join = None
            loss_out["semantic_loss"] = (semantic_loss, n_kept)
        else:
            semantic_loss = self.semantic_criterion(semantic_scores, semantic_labels)
        if self.semantic_dice and not fused:
            keep = semantic_labels != self.ignore_label
            if indexed:
                semantic_scores = F.softmax(semantic_scores[keep], dim=-1)
                one_hot = F.one_hot(semantic_labels[keep], num_classes=self.semantic_class_num)
            else:       # dropped rows -> zero rows: every column sum of the dice terms is unchanged
                w = keep.unsqueeze(1).to(semantic_scores.dtype)
                semantic_scores = F.softmax(semantic_scores, dim=-1) * w
                one_hot = F.one_hot(semantic_labels.clamp(min=0), num_classes=self.semantic_class_num) * w
            semantic_loss = semantic_loss + dice_loss_multi_classes(semantic_scores, one_hot).mean()
        if not fused:
            loss_out["semantic_loss"] = (semantic_loss, semantic_scores.sum())

        joint = epoch > self.joint_training_epoch
        if joint:
            sp_sem_labels, sp_ins_labels = loss_inp["superpoint_labels"]
            # the validity mask and its count: only the unfused branches read them (the fused kernels take the two label
            # tensors), so they are formed on first use -- four launches the default device path never needs
            _mask = []

            def _valid():
                if not _mask:
                    m = (sp_ins_labels != self.ignore_label) & (sp_sem_labels != self.ignore_label)
                    _mask.extend([m, m.sum()])
                return _mask

            sp_semantic_scores = loss_inp["sp_semantic"]
            if (not indexed and sp_semantic_scores.is_cuda and sp_semantic_scores.dim() == 2
                    and sp_semantic_scores.shape[1] <= 32 and sp_semantic_scores.shape[0] >= 1 and os.environ.get("WSIS_FUSE_SP_CE", "1") != "0"):
                import wsis_ops      # cross entropy + the logged sum of the scores: one launch each way (csrc/loss.hip)
                superpoint_semantic_loss, sp_score_sum = wsis_ops.superpoint_cross_entropy(
                    sp_semantic_scores, sp_sem_labels, self.ignore_label)
            else:
                superpoint_semantic_loss = self.superpoint_semantic_criterion(sp_semantic_scores, sp_sem_labels)
                sp_score_sum = sp_semantic_scores.sum()
            loss_out["superpoint_semantic_loss"] = (superpoint_semantic_loss, sp_score_sum)

            fused_reg = (not indexed and self.supervise_sp_offset and self.supervise_instance_size
                         and loss_inp["sp_offset_vector"][0].is_cuda
                         and os.environ.get("WSIS_FUSE_SP_LOSS", "1") != "0")
            if fused_reg:   # offset L1 + cosine, occupancy and size L1 in one launch (csrc/loss.hip)
                import wsis_ops
                pred_off, gt_off = loss_inp["sp_offset_vector"]
                pred_occ, gt_occ = loss_inp["sp_occupancy"]
                pred_size, gt_size = loss_inp["sp_instance_size"]
                offset_norm_loss, offset_dir_loss, occupancy_loss, instance_size_loss, n_reg = \
                    wsis_ops.sp_regression_losses(pred_off, gt_off, pred_occ, gt_occ, pred_size, gt_size,
                                                  sp_sem_labels, sp_ins_labels, self.ignore_label)
                loss_out["offset_norm_loss"] = (offset_norm_loss, n_reg)
                loss_out["offset_dir_loss"] = (offset_dir_loss, n_reg)

            if self.supervise_sp_offset and not fused_reg:
                pred_off, gt_off = loss_inp["sp_offset_vector"]
                pt_dist = torch.sum(torch.abs(pred_off - gt_off), dim=-1)
                sp_valid, n_valid = _valid()
                offset_norm_loss = torch.sum(pt_dist * sp_valid) / (n_valid + 1e-6)
                gt_dir = gt_off / (torch.norm(gt_off, p=2, dim=1).unsqueeze(-1) + 1e-8)
                pt_dir = pred_off / (torch.norm(pred_off, p=2, dim=1).unsqueeze(-1) + 1e-8)
                direction_diff = -(gt_dir * pt_dir).sum(-1)
                offset_dir_loss = torch.sum(direction_diff * sp_valid) / (n_valid + 1e-6)
                loss_out["offset_norm_loss"] = (offset_norm_loss, n_valid)
                loss_out["offset_dir_loss"] = (offset_dir_loss, n_valid)

            feats, sp_batch_offsets = loss_inp["sp_discriminative_features"]
            offs = [int(o) for o in sp_batch_offsets]
            slots = loss_inp.get("sp_instance_slots")      # host-known bound of the instance ids per scene (optional)
            d_losses = []
            for i in range(1, len(offs)):
                b, e = offs[i - 1], offs[i]
                if indexed:
                    valid = _valid()[0][b:e]
                    d_loss, _, _, _ = self.discriminative_loss(feats[b:e][valid], sp_ins_labels[b:e][valid])
                elif (slots is not None and feats.is_cuda and 1 <= int(slots[i - 1]) <= 64 and 1 <= e - b <= 4096
                      and self.discriminative_feature_dim == 7 and os.environ.get("WSIS_FUSE_DISC_LOSS", "1") != "0"):
                    import wsis_ops          # one launch each way (csrc/loss.hip)
                    d_loss = wsis_ops.discriminative_loss(feats[b:e], sp_ins_labels[b:e], sp_sem_labels[b:e],
                                                          int(slots[i - 1]), self.ignore_label, self.delta_v,
                                                          self.delta_d, self.param_var, self.param_dist, self.param_reg)
                elif slots is not None and 1 <= int(slots[i - 1]) <= 512:
                    d_loss = self.discriminative_loss_slots(feats[b:e], sp_ins_labels[b:e], _valid()[0][b:e],
                                                            int(slots[i - 1]))
                else:
                    d_loss = self.discriminative_loss_masked(feats[b:e], sp_ins_labels[b:e], _valid()[0][b:e])
                d_losses.append(d_loss.view(-1))
            # mean over the scenes; one scene: the mean of one value is that value (x / 1 is exact)
            sp_d_loss = torch.mean(torch.cat(d_losses)) if len(d_losses) != 1 else d_losses[0].reshape(())
            loss_out["superpoint_discriminative_loss"] = (sp_d_loss, feats.shape[0])

            if fused_reg:
                loss_out["occupancy_loss"] = (occupancy_loss, n_reg)
                loss_out["instance_size_loss"] = (instance_size_loss, n_reg)
            elif self.supervise_instance_size:
                pred_occ, gt_occ = loss_inp["sp_occupancy"]
                pred_size, gt_size = loss_inp["sp_instance_size"]
                sp_valid, n_valid = _valid()
                if indexed:
                    occupancy_loss = self.occupany_L1loss(pred_occ[sp_valid], gt_occ[sp_valid])
                    instance_size_loss = self.instance_size_L1loss(pred_size[sp_valid], gt_size[sp_valid])
                else:
                    occupancy_loss = _masked_l1(pred_occ, gt_occ, sp_valid)
                    instance_size_loss = _masked_l1(pred_size, gt_size, sp_valid)
                loss_out["occupancy_loss"] = (occupancy_loss, n_valid)
                loss_out["instance_size_loss"] = (instance_size_loss, n_valid)

        # losses_3D_WSIS.py:130-151: loss = 0.0 + 1.0 * term + ...  All weights are 1.0 there, and 0.0 + x and 1.0 * x are
        # exact in floating point: the same sum in the same order without the seven scalar multiplications (each one a
        # launch forward and one backward)
        if fused and loss_join is not None:       # the point term joins the others here
            main, side, made = loss_join
            main.wait_stream(side)
            for t in made:
                t.record_stream(main)
        # (the terms and the pairs of the reference's expression; summed left to right)
        terms, paired = [semantic_loss], 0
        self._log("point semantic loss", semantic_loss)
        if joint:
            terms.append(superpoint_semantic_loss)
            self._log("sp semantic loss", superpoint_semantic_loss)
            if self.supervise_sp_offset:
                paired |= 1 << len(terms)                  # loss + (offset_norm_loss + offset_dir_loss)
                terms += [offset_norm_loss, offset_dir_loss]
                self._log("sp offset norm loss", offset_norm_loss)
                self._log("sp offset dir loss", offset_dir_loss)
            terms.append(sp_d_loss)
            self._log("sp discriminative loss", sp_d_loss)
            if self.supervise_instance_size:
                terms += [occupancy_loss, instance_size_loss]
                self._log("sp occupancy loss", occupancy_loss)
                self._log("sp instance size loss", instance_size_loss)
        if (len(terms) > 1 and len(terms) <= 8 and all(torch.is_tensor(t) and t.is_cuda and t.numel() == 1 for t in terms)
                and os.environ.get("WSIS_FUSE_LOSS_SUM", "1") != "0"):
            import wsis_ops
            loss = wsis_ops.loss_sum(terms, paired)      # one launch instead of one per `loss = loss + term`
        else:
            loss = terms[0]
            i = 1
            while i < len(terms):
                if (paired >> i) & 1:
                    loss = loss + (terms[i] + terms[i + 1])
                    i += 2
                else:
                    loss = loss + terms[i]
                    i += 1
        return loss, loss_out

    def discriminative_loss(self, prediction, correct_label):
        """pull (delta_v) / push (L1 cdist, delta_d) / reg terms over the instances of one scene
        (losses_3D_WSIS.py:157-230)."""
        dev = prediction.device
        pred = torch.reshape(prediction, [-1, self.discriminative_feature_dim])
        unique_labels, unique_id, counts = torch.unique(correct_label, sorted=False, return_inverse=True,
                                                        return_counts=True)
        counts = counts.float()
        n = unique_labels.size(0)
        seg_sum = torch.zeros(n, self.discriminative_feature_dim, device=dev).index_add_(0, unique_id, pred)
        mu = seg_sum / counts.reshape(-1, 1)
        dist = torch.norm(pred - mu[unique_id], p=2, dim=1)
        dist = torch.square(torch.clamp(dist - self.delta_v, min=0.))
        l_var = torch.zeros(n, device=dev).index_add_(0, unique_id, dist)
        l_var = torch.sum(l_var / counts) / n
        if n <= 1:
            l_dist = torch.tensor(0., device=dev)
        else:
            d = 2. * self.delta_d - torch.cdist(mu, mu, p=1)
            d = d - torch.diagflat(torch.diag(d, 0))
            l_dist = torch.sum(torch.square(torch.clamp(d, min=0.))) / (n * (n - 1))
        l_reg = torch.sum(torch.norm(mu, p=2, dim=1))
        l_var = self.param_var * l_var
        l_dist = self.param_dist * l_dist
        l_reg = self.param_reg * l_reg
        return l_var + l_dist + l_reg, l_var, l_dist, l_reg

    def discriminative_loss_slots(self, prediction, label, valid, n_slots):
        """the same terms with the instances in ``n_slots`` fixed slots (slot = instance id; ``n_slots`` is a bound the
        HOST knows from the batch, so no ``unique`` and no sync): a one-hot [S, n_slots] membership matrix replaces
        the [S, S] same-instance matrix of ``discriminative_loss_masked`` -- instance means by one small GEMM, the
        push term on [n_slots, n_slots] instead of [S, S].  Empty slots carry weight 0.  Deterministic (no atomics)."""
        pred = torch.reshape(prediction, [-1, self.discriminative_feature_dim])
        slot = torch.arange(n_slots, device=pred.device, dtype=label.dtype)
        oh = ((label.unsqueeze(1) == slot.unsqueeze(0)) & valid.unsqueeze(1)).to(pred.dtype)      # [S, I]
        cnt = oh.sum(0)                                          # members per slot
        present = (cnt > 0).to(pred.dtype)
        n = present.sum()                                        # number of instances (0-dim tensor, no sync)
        inv = 1.0 / cnt.clamp(min=1.0)
        mu = (oh.t() @ pred) * inv.unsqueeze(1)                  # [I, D] instance means (0 for empty slots)
        mu_rows = oh @ mu                                        # [S, D] the mean of the row's instance
        dist = torch.norm(pred - mu_rows, p=2, dim=1)
        dist = torch.square(torch.clamp(dist - self.delta_v, min=0.))
        w = oh @ inv                                             # [S] 1 / size of the row's instance, 0 if dropped
        l_var = torch.sum(dist * w) / n
        l1 = (mu.unsqueeze(0) - mu.unsqueeze(1)).abs().sum(-1)   # [I, I]
        d = 2. * self.delta_d - l1
        pair = present.unsqueeze(0) * present.unsqueeze(1) * (1.0 - torch.eye(n_slots, device=pred.device, dtype=pred.dtype))
        l_dist = torch.sum(torch.square(torch.clamp(d, min=0.)) * pair) / torch.clamp(n * (n - 1), min=1.0)
        l_reg = torch.sum(torch.norm(mu, p=2, dim=1) * present)
        return self.param_var * l_var + self.param_dist * l_dist + self.param_reg * l_reg

    def discriminative_loss_masked(self, prediction, label, valid):
        """the same pull / push / reg terms (losses_3D_WSIS.py:157-230) without ``unique`` / mask indexing: rows with
        ``valid`` False get weight 0; same[i,j] = both valid and the same instance; count_i = size of i's
        instance; an instance-level sum over c becomes a row-level sum weighted by 1/count_i."""
        pred = torch.reshape(prediction, [-1, self.discriminative_feature_dim])
        v = valid.to(pred.dtype)
        same = ((label.unsqueeze(0) == label.unsqueeze(1)) & valid.unsqueeze(0) & valid.unsqueeze(1)).to(pred.dtype)
        count = same.sum(1).clamp(min=1.0)                       # [S] (1 for invalid rows: their weight is 0 anyway)
        w = v / count                                            # sums to 1 over each instance
        n = torch.round(w.sum())                                 # number of instances (0-dim tensor, no sync)
        mu = (same @ pred) / count.unsqueeze(1)                  # [S,D]: the mean of the row's instance
        dist = torch.norm(pred - mu, p=2, dim=1)
        dist = torch.square(torch.clamp(dist - self.delta_v, min=0.))
        l_var = torch.sum(dist * w) / n
        # L1 distance of every row pair (row pair (i,j) stands for instance pair (c_i,c_j)).  Broadcast form:
        # torch.cdist(p=1) took 0.85 ms forward + 0.52 ms backward for 1190 rows (one thread per pair, 7-term loop)
        l1 = (mu.unsqueeze(0) - mu.unsqueeze(1)).abs().sum(-1)
        d = 2. * self.delta_d - l1
        other = (1.0 - same) * v.unsqueeze(0) * v.unsqueeze(1)   # valid rows of DIFFERENT instances
        pair_w = other * w.unsqueeze(0) * w.unsqueeze(1)         # every ordered instance pair weighs 1 in total
        l_dist = torch.sum(torch.square(torch.clamp(d, min=0.)) * pair_w) / torch.clamp(n * (n - 1), min=1.0)
        l_reg = torch.sum(torch.norm(mu, p=2, dim=1) * w)
        return self.param_var * l_var + self.param_dist * l_dist + self.param_reg * l_reg


def _masked_l1(pred, target, row_valid):
    """nn.L1Loss()(pred[row_valid], target[row_valid]) without the boolean indexing"""
    w = row_valid.to(pred.dtype)
    while w.dim() < pred.dim():
        w = w.unsqueeze(-1)
    per_row = pred[0].numel() if pred.dim() > 1 else 1
    return torch.sum(torch.abs(pred - target) * w) / (row_valid.sum() * per_row)


def dice_loss_multi_classes(input, target, epsilon=1e-5, weight=None):
    """per-class dice on [N, nClass] probabilities / one-hot targets (losses_3D_WSIS.py:233-253)"""
    assert input.size() == target.size()
    input = input.transpose(0, 1)
    target = target.transpose(0, 1).float()
    dice = (2 * torch.sum(input * target, dim=1) + epsilon) / \
           (torch.sum(input * input, dim=1) + torch.sum(target * target, dim=1) + 1e-4 + epsilon)
    return 1. - dice

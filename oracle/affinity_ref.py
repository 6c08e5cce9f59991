"""Oracle for the inter-superpoint affinity (a16, a17).  TEST INFRASTRUCTURE ONLY.

a16 follows modules/model/backbone_3D_WSIS.py:218-244 line by line (torch, autograd gives the backward).
a17 follows train_scannetv2.py:562-570 and modules/datasets/scannetv2_dataset.py:679-736 (numpy float64);
pinned against the imported reference method via tests/golden/propagation_golden.npz."""
import numpy as np
import torch

from .scatter_ref import scatter


def edge_affinity(q, k, v, pos_enc, edge_u, edge_v):
    """backbone_3D_WSIS.py:218-244 -> (edge_affinity [E], res [max(u)+1, D])"""
    affinity = (q[edge_u] * k[edge_v]).sum(dim=1)
    affinity = affinity / np.sqrt(k.size(-1))
    affinity = affinity * pos_enc
    _max = scatter(affinity, edge_u, dim=0, reduce="max")
    affinity = affinity - _max[edge_u]
    exp_affinity = torch.exp(affinity)
    total_exp = scatter(exp_affinity, edge_u, dim=0, reduce="sum")[edge_u]
    affinity = exp_affinity / total_exp
    res = scatter(affinity.reshape(-1, 1) * v[edge_v], edge_u, dim=0, reduce="sum")
    return affinity, res


def affinity_matrix(edge_u, edge_v, edge_affinity_vals, S):
    """train_scannetv2.py:567-570"""
    A = np.zeros((S, S))
    for u, v, aff in zip(edge_u, edge_v, edge_affinity_vals):
        A[u][v] = aff
    return A


def weak_label_propagation(affinity, adjacency, sp_semantic_value, superpoint_pred_semantic,
                           superpoint_semantic_label, iterations_num, class_num):
    """scannetv2_dataset.py:679-736 -> (pseudo_label_final [S], pseudo_label_scores [S], per-class dict)"""
    S = affinity.shape[0]
    label = np.asarray(superpoint_semantic_label)
    adjacency_matrix = np.array(adjacency, dtype=np.float64) + np.eye(S)
    scores_list, pseudo_label_list, per_class = [], [], {}
    for i in range(class_num):
        if (label == i).sum() == 0:
            continue
        sem = np.zeros(adjacency_matrix.shape)
        m = (superpoint_pred_semantic == i) & (sp_semantic_value > 0.7)
        sem[m] = m.astype("int")
        for _ind, flag in enumerate(label == i):
            if flag:
                sem[_ind][_ind] = 1
        weight_matrix = affinity * adjacency_matrix * sem
        d_matrix = np.sum(weight_matrix, axis=1, keepdims=True)
        d_matrix[d_matrix == 0] += 1
        trans_matrix = weight_matrix / d_matrix
        t = trans_matrix
        for _ in range(iterations_num):
            trans_matrix = np.dot(trans_matrix, t)
        instance_prob = np.zeros(trans_matrix.shape)
        instance_prob[label == i] = trans_matrix[label == i]
        scores = np.max(instance_prob, axis=0)
        pseudo = np.argmax(instance_prob, axis=0)
        scores_list.append(scores)
        pseudo_label_list.append(pseudo)
        per_class[i] = (scores, pseudo, t)
    scores_list = np.array(scores_list)
    pseudo_label_list = np.array(pseudo_label_list)
    _ind = np.argmax(scores_list, axis=0)
    pseudo_label = np.choose(_ind, pseudo_label_list)
    pseudo_label_scores = np.choose(_ind, scores_list)
    final = np.ones(S) * -100
    unknown = (pseudo_label_scores != 0) & (label == -100)
    final[unknown] = pseudo_label[unknown]
    return final, pseudo_label_scores, per_class

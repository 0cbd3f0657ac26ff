"""name -> loop class (seesaw/loops/registry.py:7-37)."""


def build_loop_from_params(gdm, q, params):
    from .active_search import ActiveSearch, LKNNSearch
    from .graph_based import KnnProp2
    from .log_reg import LogReg2
    from .multi_reg import MultiReg
    from .multi_reg_neg import MultiRegNeg
    from .point_based import Plain
    from .pseudo_lr import PseudoLR
    from .random_results import RandomResults
    from .rocchio_update import RocchioUpdate

    cls_dict = {
        "knn_prop2": KnnProp2,
        "plain": Plain,
        "log_reg2": LogReg2,
        "pseudo_lr": PseudoLR,
        "multi_reg": MultiReg,
        "multi_reg_neg": MultiRegNeg,
        "rocchio_update": RocchioUpdate,
        "random": RandomResults,
        "active_search": ActiveSearch,
        "lknn": LKNNSearch,
    }
    cls = cls_dict.get(params.interactive)
    if cls is None:
        raise KeyError(f"loop '{params.interactive}' is not part of the accelerated path; have {sorted(cls_dict)}")
    return cls.from_params(gdm, q, params)

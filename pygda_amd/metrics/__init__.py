from .metrics import (eval_average_precision, eval_macro_f1, eval_micro_f1, eval_precision_at_k,
                      eval_recall_at_k, eval_roc_auc)

__all__ = ["eval_average_precision", "eval_macro_f1", "eval_micro_f1", "eval_precision_at_k",
           "eval_recall_at_k", "eval_roc_auc"]

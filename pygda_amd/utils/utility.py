"""Per-epoch progress line in the reference's format (pygda/utils/utility.py:3-116)."""


def logger(epoch=0, loss=0, source_train_acc=None, source_val_acc=None, target=None, time=None,
           verbose=0, train=True):
    if verbose <= 0:
        return
    parts = ["Epoch {:04d}: ".format(epoch) if train else "Test: "]
    if isinstance(loss, tuple):
        parts.append("Loss I {:.4f} | Loss O {:.4f} | ".format(loss[0], loss[1]))
    else:
        parts.append("loss {:.4f}, ".format(loss))
    if verbose > 1:
        if source_train_acc is not None:
            parts.append("source acc {:.4f}, ".format(source_train_acc))
        if target is not None:
            parts.append("target acc {:.4f}, ".format(target))
        # verbose > 2 in the reference dereferences undefined names (:96-107); nothing to mirror
        if time is not None:
            parts.append("time {:.2f}".format(time))
    print("".join(parts))

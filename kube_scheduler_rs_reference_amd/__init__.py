"""MI355X-native batched pods x nodes predicate evaluator (drop-in for the predicate
filter-and-pick path of acrlabs/kube-scheduler-rs-reference).  See DESIGN.md."""
from . import _lib  # noqa: F401
from ._lib import (FIT, SEL, TAINT, PICK_SAMPLED, PICK_BESTFIT, WANT_FIT_MASK, SEL_NEVER, KschedError)  # noqa: F401
from .evaluator import Evaluator, EvalResult, mask_words, pack_mask, unpack_mask  # noqa: F401

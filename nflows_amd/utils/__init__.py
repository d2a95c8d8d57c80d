from .torchutils import (create_alternating_binary_mask, create_mid_split_binary_mask,
                         create_random_binary_mask, merge_leading_dims, repeat_rows,
                         split_leading_dim, sum_except_batch, searchsorted)
from . import typechecks

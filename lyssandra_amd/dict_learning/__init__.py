from .ksvd import approx_ksvd, ksvd, nn_ksvd, ksvd_dict_learn, ksvd_coder  # noqa: F401
from .online_dict_learn import online_dict_learn, online_dictionary_coder  # noqa: F401
from .utils import init_dictionary, approx_error, average_mutual_coherence, force_mi  # noqa: F401

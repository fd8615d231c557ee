"""pandelos_amd — MI355X-native PanDelos hot path (k-mer dictionary matching + all-vs-all gene scoring).

The product is the HIP library behind ``include/pandelos_amd.h`` (``pandelos_amd/csrc``); this
package is the host-side mirror of the reference's Java classes for that path
(``ig/infoasys/cli/pangenes``): ``PangeneIData``, ``PangeneNative``, ``Scores``.
"""
from .pangene_idata import PangeneIData      # noqa: F401
from .scores import Scores                   # noqa: F401

__all__ = ["PangeneIData", "Scores", "PangeneNative"]


def __getattr__(name):
    if name == "PangeneNative":              # needs the built library; import lazily
        from .pangene_native import PangeneNative
        return PangeneNative
    raise AttributeError(name)

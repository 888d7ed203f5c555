"""Optional process-wide alias: make the reference's own import lines (`from vilbert.vilbert import ...`, `from lily import
Lily`, `from vilbert.vilbert_init import get_optimization`) resolve to the MI355X implementation.  Call `install()` before
the reference modules are imported."""
import sys
import types


def install() -> None:
    from . import lily, optimization, vilbert, vilbert_init
    pkg = types.ModuleType("vilbert")
    pkg.__path__ = []
    pkg.vilbert, pkg.optimization, pkg.vilbert_init = vilbert, optimization, vilbert_init
    sys.modules.update({"vilbert": pkg, "vilbert.vilbert": vilbert, "vilbert.optimization": optimization,
                        "vilbert.vilbert_init": vilbert_init, "lily": lily})

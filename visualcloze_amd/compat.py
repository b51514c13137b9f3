"""Zero-edit drop-in: make the reference's OWN import statements resolve to the MI355X path.

The reference reaches the hot path through two imports:

    models/util.py:11      from models.model import Flux, FluxLoraWrapper, FluxParams     (load_flow_model builds the module, B3)
    models/sampling.py:12  from .model import Flux
    visualcloze.py:12      from transport import Sampler, create_transport                (B2; train.py:55 too)

`install()` registers modules of exactly those names in `sys.modules` BEFORE the reference imports them, so
`visualcloze.py`, `sample.py` and `models/util.py::load_flow_model` run UNCHANGED: `load_flow_model` constructs
`visualcloze_amd.model.FluxLoraWrapper` from its own `configs[name].params` (our `FluxParams` has the same fields), loads the
same checkpoints into the same state-dict keys, and `self.sampler.sample_ode(...)(x, self.model.forward, kwargs)` lands in the
fused hipGraph loop.  Nothing else of the `models` / `transport` packages is shadowed (`models.util`, `models.sampling`,
`models.modules.*` stay the reference's).

    import visualcloze_amd.compat as compat; compat.install()      # then: from visualcloze import VisualClozeModel
    python -m visualcloze_amd.compat sample.py --model_path ...      # the same for a script, no source edit at all
"""
from __future__ import annotations

import sys
import types

_INSTALLED = {}


def install(force: bool = False) -> None:
    """Alias `models.model` and `transport` to the MI355X implementations.  Raises if the reference's own modules of those
    names were imported already (their classes would be in use) unless `force`."""
    from . import model as _model
    from . import transport as _transport
    for name in ("models.model", "transport"):
        have = sys.modules.get(name)
        if have is not None and have is not _INSTALLED.get(name) and not force:
            raise RuntimeError(f"visualcloze_amd.compat.install(): '{name}' is already imported from {getattr(have, '__file__', '?')}; "
                               "call install() before importing visualcloze / models.util (or pass force=True)")
    mm = types.ModuleType("models.model")
    mm.__doc__ = "alias of visualcloze_amd.model (visualcloze_amd.compat.install)"
    for n in ("Flux", "FluxLoraWrapper", "FluxParams", "FLUX_DEV_FILL"):
        setattr(mm, n, getattr(_model, n))
    mm.__all__ = ["Flux", "FluxLoraWrapper", "FluxParams"]
    tr = types.ModuleType("transport")
    tr.__doc__ = "alias of visualcloze_amd.transport (visualcloze_amd.compat.install)"
    for n in ("Sampler", "Transport", "create_transport"):
        setattr(tr, n, getattr(_transport, n))
    tr.__all__ = ["Sampler", "Transport", "create_transport"]
    sys.modules["models.model"], sys.modules["transport"] = mm, tr
    _INSTALLED.update({"models.model": mm, "transport": tr})
    pkg = sys.modules.get("models")            # a `models` package that is imported already sees the alias as its attribute too
    if pkg is not None:
        setattr(pkg, "model", mm)


def uninstall() -> None:
    for name, mod in list(_INSTALLED.items()):
        if sys.modules.get(name) is mod:
            del sys.modules[name]
        _INSTALLED.pop(name)


def main(argv=None) -> None:
    """`python -m visualcloze_amd.compat script.py [args...]`: install(), then run the script as __main__ (its directory first on
    sys.path, as `python script.py` would have it)."""
    import os
    import runpy
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        raise SystemExit("usage: python -m visualcloze_amd.compat <script.py> [args...]")
    install()
    script = os.path.abspath(argv[0])
    sys.argv = [script] + argv[1:]
    sys.path.insert(0, os.path.dirname(script))
    runpy.run_path(script, run_name="__main__")


if __name__ == "__main__":
    main()

"""Debug-only host switches.  The shipped configuration has none: the A/B knobs of the host side (how the launches of
an iteration are issued, on how many streams, the optimizer's chunking) are read from the environment ONLY when
``DVAE_DEBUG=1`` is set; otherwise every one of them keeps its measured default (tools/README.md lists them).
Configuration that is not an A/B switch stays a plain variable: DVAE_HIP_LIB, DVAE_RCCL_LIB, DVAE_COMM."""
import os

def enabled():
    return os.environ.get("DVAE_DEBUG", "0") == "1"


_WARNED = set()


def knob(name, default):
    """Value of the debug switch `name` (a string) when DVAE_DEBUG=1, else `default` -- with a one-time warning when the
    variable is set but ignored (a user who exported e.g. DVAE_REPLAY=eager to work around a problem should learn that it
    only acts together with DVAE_DEBUG=1)."""
    if enabled():
        return os.environ.get(name, default)
    if name in os.environ and name not in _WARNED:
        _WARNED.add(name)
        import warnings
        warnings.warn("%s=%s is ignored: the host-side debug switches are read only when DVAE_DEBUG=1 is set"
                      % (name, os.environ[name]), RuntimeWarning, stacklevel=2)
    return default

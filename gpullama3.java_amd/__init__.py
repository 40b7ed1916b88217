"""gpullama3.java_amd — MI355X-native transformer forward pass for GPULlama3.java.

Host-side mirror (Python, because no JDK exists in this image) of the reference's plan interface
(J/tornadovm/TornadoVMMasterPlan.java:30-85) over the C-ABI library ``libgpullama_hip.so``
(include/gpullama3_hip.h).  The directory name contains a dot, so import it through
``__graft_entry__.load_package()`` which registers it as ``gpullama3_java_amd``.

The HIP library is loaded lazily by ``hip.lib()`` and fails loudly if it is missing: there is no
CPU fallback in this package.
"""
from . import gguf, javarand, synth  # noqa: F401  (host-side, no GPU needed)

__all__ = ["gguf", "javarand", "synth"]

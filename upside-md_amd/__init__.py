"""upside-md_amd: MI355X-native implementation of Upside's MD inner loop (force pass + integrator)
behind the reference's engine_c_library C-ABI.  The directory name carries a hyphen, so load it with
`__graft_entry__.load_package()` (registers it as module `upside_md_amd`)."""
from . import h5lite, config, engine, replicas  # noqa: F401
from .engine import Upside, UpsideLibrary, default_library, PRODUCT_LIB  # noqa: F401

"""`seesaw` -- the reference's import path, served by this repository's implementation.

A caller written against orm011/seesaw (`from seesaw.query_interface import InteractiveQuery`,
`from seesaw.indices.multiscale.multiscale_index import MultiscaleIndex`, `import seesaw.seesaw_bench`, an index
directory whose info.json says "seesaw.indices.multiscale.multiscale_index.MultiscaleIndex") resolves to the
MI355X implementation in `seesaw_amd` without an import rewrite: every `seesaw.X` is the very module object
`seesaw_amd.X` (one set of classes, `isinstance` holds across the two names).  Nothing of the reference is in here;
modules it has and this repository does not (web UI, Ray actors, ...) raise ModuleNotFoundError as usual.

Keep this directory out of sys.path's way when the REAL reference must be imported (oracle/_ref_import.py puts
/root/reference first)."""
import importlib
import importlib.abc
import importlib.util
import sys

import seesaw_amd

_PREFIX, _TARGET = __name__ + ".", seesaw_amd.__name__ + "."


class _AliasLoader(importlib.abc.Loader):
    def __init__(self, target):
        self.target = target

    def create_module(self, spec):
        return importlib.import_module(self.target)  # the seesaw_amd module itself

    def exec_module(self, module):
        pass  # already executed under its own name


class _AliasFinder(importlib.abc.MetaPathFinder):
    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(_PREFIX):
            return None
        real = _TARGET + fullname[len(_PREFIX):]
        try:
            found = importlib.util.find_spec(real)
        except ModuleNotFoundError:
            return None
        if found is None:
            return None
        return importlib.util.spec_from_loader(fullname, _AliasLoader(real), is_package=found.submodule_search_locations is not None)


if not any(isinstance(f, _AliasFinder) for f in sys.meta_path):
    sys.meta_path.insert(0, _AliasFinder())

__path__ = []  # submodules come from the finder above, never from this directory
__version__ = getattr(seesaw_amd, "__version__", "0")

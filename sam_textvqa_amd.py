"""Import alias: the package directory is `sam-textvqa_amd/` (a hyphen is not a legal Python
identifier), so `import sam_textvqa_amd` loads that directory as a package."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "sam-textvqa_amd")]
__file__ = _os.path.join(__path__[0], "__init__.py")
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))

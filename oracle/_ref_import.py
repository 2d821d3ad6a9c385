"""Import harness for the upstream reference (THIS CONTAINER ONLY).

TEST INFRASTRUCTURE - never imported by the product path (realise_amd/).

Puts /root/reference on sys.path (so the vendored transformers 2.2.2 shadows
site-packages) and stubs the six third-party modules the reference imports but
never touches on the hot path.  Used only by
oracle/make_golden.py and by the CPU tests that pin the restatement when
/root/reference is present.  Nothing here travels to the GPU box in a usable
form: /root/reference does not exist there and the callers skip.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("REALISE_REFERENCE_ROOT", "/root/reference")


def reference_available():
    return os.path.isfile(os.path.join(REFERENCE_ROOT, "src", "models.py"))


def _stub(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m


_CACHE = None


def import_reference():
    """Returns (models_module, BertConfig) of the upstream reference (imported once per process: a second import would hand out
    a BertConfig class the cached ``models`` module does not recognise)."""
    global _CACHE
    if _CACHE is not None:
        return _CACHE
    if not reference_available():
        raise RuntimeError("reference tree not present at %s" % REFERENCE_ROOT)
    sys.dont_write_bytecode = True
    _stub("torchcrf", CRF=object)
    _stub("boto3")
    _stub("botocore")
    _stub("botocore.config", Config=object)
    _stub("botocore.exceptions", ClientError=Exception)
    _stub("sacremoses")
    _stub("opencc", OpenCC=object)
    _stub("pypinyin", pinyin=None, Style=types.SimpleNamespace(TONE3=None))
    for p in (os.path.join(REFERENCE_ROOT, "src"), REFERENCE_ROOT):
        if p not in sys.path:
            sys.path.insert(0, p)
    # make sure a previously imported site-packages transformers does not win
    for k in [k for k in sys.modules if k == "transformers" or k.startswith("transformers.")]:
        del sys.modules[k]
    from transformers import BertConfig  # vendored 2.2.2
    import models
    _CACHE = (models, BertConfig)
    return _CACHE

"""Whisper language codes (data): the 99 language tokens of the multilingual vocabulary in token-id order
(<|en|> = 50259 ... <|su|> = 50357).  Used to validate `force_language` (the reference checks membership in the keys of its
language table, main.py:84,550-551) and to name `detect_language` results (`"<|xx|>"`).  Only the codes matter on this path;
the English language names of the reference's table are not needed and not carried.
"""
LANGUAGE_CODES = tuple((
    "en zh de es ru ko fr ja pt tr pl ca nl ar sv it id hi fi vi he uk el ms cs ro da hu ta no th ur hr bg "
    "lt la mi ml cy sk te fa lv bn sr az sl kn et mk br eu is hy ne mn bs kk sq sw gl mr pa si km sn yo so "
    "af oc ka be tg sd gu am yi lo uz fo ht ps tk nn mt sa lb my bo tl mg as tt haw ln ha ba jw su "
).split())
assert len(LANGUAGE_CODES) == 99 and len(set(LANGUAGE_CODES)) == 99
# membership test target of `check_language` (same truth value as `language in LANGUAGES` of the reference)
LANGUAGES = frozenset(LANGUAGE_CODES)

"""Import-only stand-in so that xenoverse/metacontrol/random_acrobot.py can be imported in the build container
(gymnasium is not installed).  It provides NO behaviour: oracle/gen_golden.py calls only the methods the reference
file itself defines (_dsdt, _terminal), which need nothing from this base class but the `book_or_nips` default."""


class AcrobotEnv(object):
    book_or_nips = "book"      # gymnasium's default
    render_mode = None

/* stand-in: the hot-path translation units need nothing from this third-party header */
#pragma once

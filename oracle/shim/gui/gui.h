// ORACLE stub for <gui/gui.h>: IQFrontEnd only calls gui::waterfall.setRawFFTSize (iq_frontend.cpp:305).
#pragma once
namespace gui {
    struct WaterfallStub {
        int rawFFTSize = 0;
        void setRawFFTSize(int size) { rawFFTSize = size; }
    };
    inline WaterfallStub waterfall;
}
